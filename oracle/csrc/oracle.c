/* CPU oracle, C part -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain C restatement of two float32 pieces of the hot path whose arithmetic
 * lives in third-party code absent from /root/reference (gensim 4.3.3, numpy/BLAS):
 *
 *   orc_sim_chain    gensim Similarity.__getitem__ -> MatrixSimilarity.get_similarities:
 *                    scores = index[D,K] . q[K]   (reference call sites webui.py:205,352).
 *                    BLAS sgemv fixes no summation order; this oracle *defines* it as the
 *                    k-ordered fused chain acc = fmaf(a[k], q[k], acc), k = 0..K-1, which is
 *                    exactly what gfx950's v_mfma_f32_32x32x2_f32 computes.
 *
 *   orc_d2v_infer_dm the same for dm=1 (PV-DM, sum or mean of context + document vector): see its own comment below.
 *
 *   orc_d2v_infer    gensim Doc2Vec.infer_vector for dm=0 (PV-DBOW), negative sampling,
 *                    hs=0 (reference call sites genmodel.py:159,169; webui.py:106,185):
 *                    doc2vec.py::infer_vector -> doc2vec_inner.pyx::train_document_dbow ->
 *                    fast_document_dbow_neg, with word2vec_inner's 48-bit LCG, EXP_TABLE and
 *                    bisect_left over cum_table.  PARITY UNPINNED (gensim absent; restated
 *                    from the published algorithm).  Explicit inputs replace gensim's hidden
 *                    state: the start vector v0 (gensim: SFC64(hash(' '.join(words)))) and a
 *                    per-document 64-bit seed from which each epoch's 48-bit LCG state is
 *                    derived with splitmix64 (gensim: two draws from model.random per epoch).
 *                    sdot's order is defined as: 64 lane partials p[l] = fma-chain over
 *                    elements l, l+64, l+128, ... then the xor butterfly 32,16,8,4,2,1.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; fmaf() is explicit).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EXP_TABLE_SIZE 1000
#define ORC_MAX_EXP 6

void orc_sim_chain(const float* index, int64_t D, int K, int64_t ld, const float* q, float* out) {
    for (int64_t d = 0; d < D; ++d) {
        const float* a = index + d * ld;
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) acc = fmaf(a[k], q[k], acc);
        out[d] = acc;
    }
}

/* word2vec_inner.pyx: EXP_TABLE[i] = exp((i / 1000 * 2 - 1) * 6); EXP_TABLE[i] /= (EXP_TABLE[i] + 1)  (REAL_t) */
void orc_exp_table(float* table) {
    for (int i = 0; i < ORC_EXP_TABLE_SIZE; ++i) {
        float e = (float)exp((i / (float)ORC_EXP_TABLE_SIZE * 2 - 1) * ORC_MAX_EXP);
        table[i] = (float)(e / (e + 1));
    }
}

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

/* the dot-product order shared with the HIP kernel (one wavefront per document) */
static float dot_wave64(const float* v, const float* w, int dim) {
    float p[64];
    for (int l = 0; l < 64; ++l) {
        float acc = 0.0f;
        for (int i = l; i < dim; i += 64) acc = fmaf(v[i], w[i], acc);
        p[l] = acc;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        float t[64];
        for (int l = 0; l < 64; ++l) t[l] = p[l] + p[l ^ m];
        memcpy(p, t, sizeof(p));
    }
    return p[0];
}

/* word2vec_inner.pyx bisect_left over the uint32 cumulative table */
static uint32_t bisect_left_u32(const uint32_t* a, uint64_t x, uint64_t lo, uint64_t hi) {
    while (hi > lo) {
        uint64_t mid = (lo + hi) >> 1;
        if (a[mid] >= x) hi = mid; else lo = mid + 1;
    }
    return (uint32_t)lo;
}

/* One document.  words: vocabulary indices, -1 = out of vocabulary (dropped).
 * exp_scale: (EXP_TABLE_SIZE / MAX_EXP / 2) as the compiled gensim evaluates it (83 with
 * integer folding, 83.33.. with true division) -- a parameter because it is unverifiable here. */
static void infer_one(const float* syn1neg, const uint32_t* cum_table, int64_t V, const uint32_t* sample_int,
                      int dim, const int32_t* words, int nwords, const float* v0, uint64_t seed,
                      int epochs, float alpha0, float min_alpha, int negative, double exp_scale,
                      const float* exp_table, float* out) {
    const uint64_t MOD = 281474976710655ULL;
    float* v = out;
    float* work = (float*)malloc(sizeof(float) * (size_t)dim);
    memcpy(v, v0, sizeof(float) * (size_t)dim);
    /* doc2vec.py::infer_vector: alpha_delta = (alpha - min_alpha) / max(epochs - 1, 1)  (Python floats = double) */
    double alpha = (double)alpha0;
    double alpha_delta = ((double)alpha0 - (double)min_alpha) / (double)(epochs - 1 > 1 ? epochs - 1 : 1);
    for (int e = 0; e < epochs; ++e) {
        uint64_t next_random = splitmix64(seed + (uint64_t)e) & MOD;
        const float a = (float)alpha;                       /* c.alpha is REAL_t */
        for (int i = 0; i < nwords; ++i) {
            int32_t w = words[i];
            if (w < 0 || w >= V) continue;                  /* token not in vocabulary */
            if (sample_int) {                               /* c.sample != 0 */
                uint64_t r = next_random >> 16;             /* random_int32 */
                next_random = (next_random * 25214903917ULL + 11) & MOD;
                if ((uint64_t)sample_int[w] < r) continue;
            }
            /* fast_document_dbow_neg */
            memset(work, 0, sizeof(float) * (size_t)dim);
            for (int d = 0; d < negative + 1; ++d) {
                uint32_t target;
                float label;
                if (d == 0) {
                    target = (uint32_t)w;
                    label = 1.0f;
                } else {
                    target = bisect_left_u32(cum_table, (next_random >> 16) % cum_table[V - 1], 0, (uint64_t)V);
                    next_random = (next_random * 25214903917ULL + 11) & MOD;
                    if (target == (uint32_t)w) continue;
                    label = 0.0f;
                }
                const float* row = syn1neg + (int64_t)target * dim;
                float f = dot_wave64(v, row, dim);
                if (f <= -ORC_MAX_EXP || f >= ORC_MAX_EXP) continue;
                f = exp_table[(int)((double)(f + (float)ORC_MAX_EXP) * exp_scale)];
                float g = (label - f) * a;
                for (int k = 0; k < dim; ++k) work[k] = fmaf(g, row[k], work[k]);   /* saxpy */
            }
            for (int k = 0; k < dim; ++k) v[k] = v[k] + work[k];                    /* lockf = 1.0 */
        }
        alpha -= alpha_delta;
    }
    free(work);
}

/* docs in CSR form: words of document d are words[doc_ptr[d] .. doc_ptr[d+1]) */
void orc_d2v_infer(const float* syn1neg, const uint32_t* cum_table, int64_t V, const uint32_t* sample_int,
                   int dim, const int64_t* doc_ptr, const int32_t* words, int64_t ndocs,
                   const float* v0, const uint64_t* seeds, int epochs, float alpha, float min_alpha,
                   int negative, double exp_scale, float* out) {
    float table[ORC_EXP_TABLE_SIZE];
    orc_exp_table(table);
    for (int64_t d = 0; d < ndocs; ++d) {
        infer_one(syn1neg, cum_table, V, sample_int, dim, words + doc_ptr[d], (int)(doc_ptr[d + 1] - doc_ptr[d]),
                  v0 + d * dim, seeds[d], epochs, alpha, min_alpha, negative, exp_scale, table, out + d * dim);
    }
}


/* ---------------------------------------------------------------------------------------------
 * orc_d2v_infer_dm -- gensim Doc2Vec.infer_vector for dm=1 (PV-DM, non-concatenative), negative sampling, hs=0:
 * doc2vec.py::infer_vector -> doc2vec_inner.pyx::train_document_dm (learn_doctags = 1, learn_words = learn_hidden = 0)
 * -> fast_document_dm_neg.  BASELINE.json's north_star names this form; the reference itself runs dm=0 (genmodel.py:159).
 * PARITY UNPINNED (gensim 4.3.3 absent; restated from the published algorithm).  Per epoch:
 *   1. the kept words: in vocabulary and surviving sub-sampling (one LCG draw per in-vocabulary word, as orc_d2v_infer);
 *   2. a reduced window b[i] in [0, window) per kept position (gensim draws them with ONE numpy randint call from the
 *      model's hidden RandomState; here they continue the same explicit 48-bit LCG stream: b[i] = (state >> 16) % window,
 *      one step per position, after the pass of 1.);
 *   3. for every kept position i: context = kept positions [max(0, i - window + b), min(n, i + window + 1 - b)) without i;
 *      l1 = sum of the context words' vectors (ascending position) + the document vector, count = context size + 1,
 *      dm_mean: l1 *= 1 / count;  work = sum over the (1 + negative) targets of g * syn1neg[target] with
 *      f = l1 . syn1neg[target] (targets, draws, EXP_TABLE and skips exactly as fast_document_dbow_neg above);
 *      not dm_mean: work *= 1 / count;  document vector += work (lockf = 1).
 * word_vectors: float32 [V][dim], the model's wv.vectors (frozen during inference).  The float32 operation order is the
 * one shared with the HIP kernel: sums element-wise in the order given, dot = dot_wave64.  Documents with more than
 * max_kept kept words are an error for the caller to rule out (the HIP kernel holds a document's kept list in LDS).
 */
static void infer_one_dm(const float* syn1neg, const float* wv, const uint32_t* cum_table, int64_t V, const uint32_t* sample_int,
                         int dim, const int32_t* words, int nwords, const float* v0, uint64_t seed, int epochs, float alpha0,
                         float min_alpha, int negative, double exp_scale, int window, int dm_mean, const float* exp_table, float* out) {
    const uint64_t MOD = 281474976710655ULL;
    float* v = out;
    float* work = (float*)malloc(sizeof(float) * (size_t)dim);
    float* l1 = (float*)malloc(sizeof(float) * (size_t)dim);
    int32_t* kept = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nwords > 0 ? nwords : 1));
    int32_t* red = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nwords > 0 ? nwords : 1));
    memcpy(v, v0, sizeof(float) * (size_t)dim);
    double alpha = (double)alpha0;
    double alpha_delta = ((double)alpha0 - (double)min_alpha) / (double)(epochs - 1 > 1 ? epochs - 1 : 1);
    for (int e = 0; e < epochs; ++e) {
        uint64_t next_random = splitmix64(seed + (uint64_t)e) & MOD;
        const float a = (float)alpha;
        int n = 0;
        for (int i = 0; i < nwords; ++i) {
            int32_t w = words[i];
            if (w < 0 || w >= V) continue;
            if (sample_int) {
                uint64_t r = next_random >> 16;
                next_random = (next_random * 25214903917ULL + 11) & MOD;
                if ((uint64_t)sample_int[w] < r) continue;
            }
            kept[n++] = w;
        }
        for (int i = 0; i < n; ++i) {
            red[i] = (int32_t)((next_random >> 16) % (uint64_t)window);
            next_random = (next_random * 25214903917ULL + 11) & MOD;
        }
        for (int i = 0; i < n; ++i) {
            int j = i - window + red[i], k = i + window + 1 - red[i];
            if (j < 0) j = 0;
            if (k > n) k = n;
            memset(l1, 0, sizeof(float) * (size_t)dim);
            float count = 0.0f;
            for (int m = j; m < k; ++m) {
                if (m == i) continue;
                count += 1.0f;
                const float* row = wv + (int64_t)kept[m] * dim;
                for (int c = 0; c < dim; ++c) l1[c] = l1[c] + row[c];
            }
            count += 1.0f;                                                   /* the one document tag */
            for (int c = 0; c < dim; ++c) l1[c] = l1[c] + v[c];
            const float inv_count = 1.0f / count;
            if (dm_mean)
                for (int c = 0; c < dim; ++c) l1[c] = l1[c] * inv_count;
            memset(work, 0, sizeof(float) * (size_t)dim);
            const int32_t w = kept[i];
            for (int d = 0; d < negative + 1; ++d) {
                uint32_t target;
                float label;
                if (d == 0) {
                    target = (uint32_t)w;
                    label = 1.0f;
                } else {
                    target = bisect_left_u32(cum_table, (next_random >> 16) % cum_table[V - 1], 0, (uint64_t)V);
                    next_random = (next_random * 25214903917ULL + 11) & MOD;
                    if (target == (uint32_t)w) continue;
                    label = 0.0f;
                }
                const float* row = syn1neg + (int64_t)target * dim;
                float f = dot_wave64(l1, row, dim);
                if (f <= -ORC_MAX_EXP || f >= ORC_MAX_EXP) continue;
                f = exp_table[(int)((double)(f + (float)ORC_MAX_EXP) * exp_scale)];
                float g = (label - f) * a;
                for (int c = 0; c < dim; ++c) work[c] = fmaf(g, row[c], work[c]);
            }
            if (!dm_mean)
                for (int c = 0; c < dim; ++c) work[c] = work[c] * inv_count;
            for (int c = 0; c < dim; ++c) v[c] = v[c] + work[c];
        }
        alpha -= alpha_delta;
    }
    free(work);
    free(l1);
    free(kept);
    free(red);
}

void orc_d2v_infer_dm(const float* syn1neg, const float* word_vectors, const uint32_t* cum_table, int64_t V, const uint32_t* sample_int,
                      int dim, const int64_t* doc_ptr, const int32_t* words, int64_t ndocs, const float* v0, const uint64_t* seeds,
                      int epochs, float alpha, float min_alpha, int negative, double exp_scale, int window, int dm_mean, float* out) {
    float table[ORC_EXP_TABLE_SIZE];
    orc_exp_table(table);
    for (int64_t d = 0; d < ndocs; ++d) {
        infer_one_dm(syn1neg, word_vectors, cum_table, V, sample_int, dim, words + doc_ptr[d], (int)(doc_ptr[d + 1] - doc_ptr[d]),
                     v0 + d * dim, seeds[d], epochs, alpha, min_alpha, negative, exp_scale, window, dm_mean, table, out + d * dim);
    }
}


/* ---------------------------------------------------------------------------------------------
 * orc_d2v_train -- gensim Doc2Vec(dm=0).train as genmodel.py:159-162 runs it (vector_size=300, window=50, min_count=1,
 * workers=1, dm=0, epochs=100; defaults negative=5, hs=0, sample=1e-3, alpha=0.025, min_alpha=1e-4, dbow_words=0).
 * PARITY UNPINNED (gensim 4.3.3 absent): restated from doc2vec.py / doc2vec_inner.pyx::train_document_dbow ->
 * fast_document_dbow_neg (learn_doctags = learn_hidden = 1, lockf = 1) and word2vec.py::_train_epoch / _get_next_alpha:
 *   - ONE worker: documents are visited in corpus order, every epoch;
 *   - jobs: consecutive documents are batched while their raw word counts fit batch_words (10000); a job's alpha is
 *     alpha0 - (alpha0 - min_alpha) * (epoch + documents_pushed_before_the_job / total_documents) / epochs, not below min_alpha;
 *   - per document: the LCG state is drawn afresh (gensim: two draws from the model's RandomState; here the pure function
 *     splitmix64(seed + epoch * ndocs + doc), the same explicit-seed device as orc_d2v_infer);
 *   - per kept word (in-vocabulary, surviving sub-sampling): fast_document_dbow_neg as in inference, PLUS the hidden-layer
 *     update syn1neg[target] += g * doc_vector (with the document vector as it was before this word's own update).
 * The float32 operation order is the one shared with the HIP kernel: dot = dot_wave64; saxpy element-wise fmaf.
 * syn1neg [V][dim] and doc_vectors [ndocs][dim] are updated in place.
 */
void orc_d2v_train(float* syn1neg, float* doc_vectors, const uint32_t* cum_table, int64_t V, const uint32_t* sample_int, int dim,
                   const int64_t* doc_ptr, const int32_t* words, int64_t ndocs, int epochs, float alpha0, float min_alpha,
                   int negative, double exp_scale, uint64_t seed, int batch_words) {
    const uint64_t MOD = 281474976710655ULL;
    float table[ORC_EXP_TABLE_SIZE];
    orc_exp_table(table);
    float* work = (float*)malloc(sizeof(float) * (size_t)dim);
    for (int e = 0; e < epochs; ++e) {
        int64_t job_first = 0;          /* first document of the current job */
        int64_t job_words = 0;
        float a = 0.0f;
        for (int64_t d = 0; d < ndocs; ++d) {
            const int64_t nw = doc_ptr[d + 1] - doc_ptr[d];
            if (d == 0 || job_words + nw > batch_words) {      /* a new job starts with this document */
                job_first = d;
                job_words = 0;
                double progress = ((double)e + (double)job_first / (double)ndocs) / (double)epochs;
                double al = (double)alpha0 - ((double)alpha0 - (double)min_alpha) * progress;
                if (al < (double)min_alpha) al = (double)min_alpha;
                a = (float)al;
            }
            job_words += nw;
            float* v = doc_vectors + d * dim;
            uint64_t next_random = splitmix64(seed + (uint64_t)e * (uint64_t)ndocs + (uint64_t)d) & MOD;
            for (int64_t i = doc_ptr[d]; i < doc_ptr[d + 1]; ++i) {
                int32_t w = words[i];
                if (w < 0 || w >= V) continue;
                if (sample_int) {
                    uint64_t r = next_random >> 16;
                    next_random = (next_random * 25214903917ULL + 11) & MOD;
                    if ((uint64_t)sample_int[w] < r) continue;
                }
                memset(work, 0, sizeof(float) * (size_t)dim);
                for (int t = 0; t < negative + 1; ++t) {
                    uint32_t target;
                    float label;
                    if (t == 0) {
                        target = (uint32_t)w;
                        label = 1.0f;
                    } else {
                        target = bisect_left_u32(cum_table, (next_random >> 16) % cum_table[V - 1], 0, (uint64_t)V);
                        next_random = (next_random * 25214903917ULL + 11) & MOD;
                        if (target == (uint32_t)w) continue;
                        label = 0.0f;
                    }
                    float* row = syn1neg + (int64_t)target * dim;
                    float f = dot_wave64(v, row, dim);
                    if (f <= -ORC_MAX_EXP || f >= ORC_MAX_EXP) continue;
                    f = table[(int)((double)(f + (float)ORC_MAX_EXP) * exp_scale)];
                    float g = (label - f) * a;
                    for (int k = 0; k < dim; ++k) work[k] = fmaf(g, row[k], work[k]);     /* saxpy: work += g * syn1neg[target] */
                    for (int k = 0; k < dim; ++k) row[k] = fmaf(g, v[k], row[k]);          /* learn_hidden: syn1neg[target] += g * doc */
                }
                for (int k = 0; k < dim; ++k) v[k] = v[k] + work[k];                       /* learn_doctags, lockf = 1 */
            }
        }
    }
    free(work);
}

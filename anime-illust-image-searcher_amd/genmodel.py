#!/usr/bin/env python3
"""python genmodel.py [--update]                              (same flag as the reference, genmodel.py:109-177)

Reads tags-wd-tagger.txt, writes tags-wd-tagger_doc2vec_idx.csv, doc2vec_dictionary, doc2vec_model, doc2vec_index
and the five BM25 pickles.  Like the reference (genmodel.py:159-162) it TRAINS the Doc2Vec model -- PV-DBOW, 300-d, 100
epochs, on the GPU (hipts_d2v_train) -- then infers every document vector and builds the BM25 statistics, also on the GPU.
--epochs N shortens training and inference (tests); --d2v-mode sequential|parallel overrides the schedule that workers=1 picks
(sequential = the reference's one-worker order, reproducible, one wavefront: small corpora; parallel = all documents of an
epoch at once).  --synthetic-d2v skips training and builds a seeded stand-in model from the corpus counts."""
import argparse
import copy
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np  # noqa: E402

TRAIN_EPOCHS = 100        # genmodel.py:15
VECTOR_LENGTH = 300       # genmodel.py:16


def main(arg_str: list) -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument('--update', action='store_true', help='add new images to index')
    parser.add_argument('--d2v-model', default='doc2vec_model')
    parser.add_argument('--synthetic-d2v', action='store_true')
    parser.add_argument('--epochs', type=int, default=TRAIN_EPOCHS)
    parser.add_argument('--d2v-mode', choices=['sequential', 'parallel'], default=None)
    parser.add_argument('--device', type=int, default=0)
    args = parser.parse_args(arg_str)
    from hiptagsearch import synth
    from hiptagsearch.bm25 import gen_and_save_bm25_index
    from hiptagsearch.d2v import Doc2Vec, Doc2VecInference
    from hiptagsearch.index import Similarity
    from hiptagsearch.textio import Dictionary, count_non_empty_lines, read_documents_and_gen_idx_text

    if args.update:                                                                  # genmodel.py:123-130
        if os.path.exists('tags-wd-tagger_doc2vec_idx.csv'):
            with open('tags-wd-tagger_doc2vec_idx.csv', 'r', encoding='utf-8') as f, \
                    open('tags-wd-tagger_doc2vec_idx.csv.bak', 'w', encoding='utf-8') as f_bak:
                f_bak.write(f.read())
        else:
            print('tags-wd-tagger_doc2vec_idx.csv not found')
            raise SystemExit(1)
    processed_docs, _ = read_documents_and_gen_idx_text('tags-wd-tagger.txt')         # :132
    processed_docs_for_bm25 = copy.deepcopy(processed_docs)
    if args.update:                                                                  # :137-148
        dictionary = pickle.load(open('doc2vec_dictionary', 'rb'))
        model = Doc2VecInference.load(args.d2v_model, args.device)
        index = Similarity.load('doc2vec_index', args.device)
        before = count_non_empty_lines('tags-wd-tagger_doc2vec_idx.csv.bak')
        print(f'update index: {len(processed_docs) - before} files')
        processed_docs = processed_docs[before:]
    else:
        dictionary = Dictionary(processed_docs)                                      # :151-156
        pickle.dump(dictionary, open('doc2vec_dictionary', 'wb'))
        if args.synthetic_d2v:
            print('--synthetic-d2v: building the seeded stand-in model (no training).')
            # vocabulary indices by descending frequency, as gensim orders wv
            counts = {}
            for d in processed_docs:
                for t in d:
                    counts[t] = counts.get(t, 0) + 1
            vocab = sorted(counts, key=lambda t: -counts[t])
            m = synth.d2v_model(np.array([counts[t] for t in vocab]), dim=300)
            model = Doc2VecInference(m['syn1neg'], m['cum_table'], m['sample_int'], {t: i for i, t in enumerate(vocab)},
                                     epochs=args.epochs, device=args.device)
            model.save(args.d2v_model)
        else:
            # gen Doc2Vec model with specified number of dimensions                              genmodel.py:158-162
            doc2vec_model = Doc2Vec(vector_size=VECTOR_LENGTH, window=50, min_count=1, workers=1, dm=0, device=args.device)
            doc2vec_model.build_vocab(processed_docs)
            doc2vec_model.train(processed_docs, total_examples=doc2vec_model.corpus_count, epochs=args.epochs, mode=args.d2v_mode)
            print(f'Doc2Vec trained: {doc2vec_model.corpus_count} documents, {len(doc2vec_model.key_to_index)} tags, '
                  f'{args.epochs} epochs, {doc2vec_model.last_mode} schedule')
            doc2vec_model.save(args.d2v_model)
            model = doc2vec_model.inference()
        index = None
    if processed_docs:
        vecs = model.infer_vectors(processed_docs)                                   # :168-169, one launch
        if index is None:
            index = Similarity('doc2vec_index', None, model.vector_size, args.device, capacity=len(vecs))
        index.add_matrix(vecs)                                                       # :170-173 (ndarray docs: stored as given)
    index.save('doc2vec_index')                                                      # :175
    gen_and_save_bm25_index(processed_docs_for_bm25, dictionary, args.device)        # :177


if __name__ == "__main__":
    main(sys.argv[1:])

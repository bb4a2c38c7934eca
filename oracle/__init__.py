"""CPU oracle for the hip-tagsearch hot path  --  TEST INFRASTRUCTURE ONLY.

A plain numpy / C / torch-CPU restatement of the reference's algorithm for the
indexing + query-scoring hot path.  Nothing in the product
package (`anime-illust-image-searcher_amd/`) may import, link or execute
anything under this directory: only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg use it, and only as the checker.

Pinning status:
  * bm25.py, tags.py, search.filter_searched_result, textio  -- PINNED: checked
    bit-for-bit against golden vectors produced by running the reference's own
    numpy code (tests/golden/make_golden.py -> tests/golden/g*.json|npz).
  * vit.py (timm ViT forward), d2v.py (gensim Doc2Vec.infer_vector),
    search.similarity (gensim Similarity.__getitem__)  -- PARITY UNPINNED: the
    arithmetic lives in third-party packages that are absent from
    /root/reference and not installed (timm 1.0.9, gensim 4.3.3, torch 2.5.1);
    the reference holds no test, fixture or known-answer vector for them.  These
    files restate the published algorithms and are anchored on the reference's
    call sites only.
"""

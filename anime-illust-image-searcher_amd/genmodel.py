#!/usr/bin/env python3
"""python genmodel.py [--update]                              (same flag as the reference, genmodel.py:109-177)

Reads tags-wd-tagger.txt, writes tags-wd-tagger_doc2vec_idx.csv, doc2vec_dictionary, doc2vec_index
and the five BM25 pickles.  Doc2Vec TRAINING (genmodel.py:159-162) is out of scope (DESIGN.md
section 6): a frozen model is loaded from --d2v-model (hiptagsearch format), or --synthetic-d2v
builds the seeded stand-in from the corpus counts.  Inference of every document vector and the BM25
statistics run on the GPU."""
import argparse
import copy
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np  # noqa: E402


def main(arg_str: list) -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument('--update', action='store_true', help='add new images to index')
    parser.add_argument('--d2v-model', default='doc2vec_model')
    parser.add_argument('--synthetic-d2v', action='store_true')
    parser.add_argument('--epochs', type=int, default=100)
    parser.add_argument('--device', type=int, default=0)
    args = parser.parse_args(arg_str)
    from hiptagsearch import synth
    from hiptagsearch.bm25 import gen_and_save_bm25_index
    from hiptagsearch.d2v import Doc2VecInference
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
        if args.synthetic_d2v or not os.path.exists(args.d2v_model):
            print('No trained Doc2Vec model: building the seeded synthetic stand-in (training is out of scope).')
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
            model = Doc2VecInference.load(args.d2v_model, args.device)
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

#!/usr/bin/env python3
"""python query.py "1girl blue_eyes:+2 hat:-1" [--topn 50]     -- the webui.py query function
(find_similar_documents, webui.py:345) without the Streamlit UI."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("query")
    ap.add_argument("--topn", type=int, default=50)
    ap.add_argument("--compat-rerank", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    from hiptagsearch import search
    eng = search.load_engine(a.device, compat_rerank=a.compat_rerank)
    search.set_engine(eng)
    for doc_id, score in search.find_similar_documents(a.query, a.topn):
        print("%.6f\t%s" % (score, eng.image_files_name_tags_arr[doc_id].split(",")[0]))


if __name__ == "__main__":
    main(sys.argv[1:])

"""Text formats and the tag dictionary (host logic).

  read_documents_and_gen_idx_text   genmodel.py:21-43   (same name / behaviour / side file)
  Dictionary                        gensim.corpora.Dictionary as genmodel.py:151 builds it: ids are
                                    assigned document by document to the *sorted* new tokens of each
                                    document [published gensim behaviour]; pickled as a plain object
                                    with a `token2id` dict, which is all webui.py:364-371 reads.
"""
from typing import Dict, Iterable, List, Tuple


class Dictionary:
    def __init__(self, documents: Iterable[Iterable[str]] = None):
        self.token2id: Dict[str, int] = {}
        self.dfs: Dict[int, int] = {}
        self.num_docs = 0
        if documents is not None:
            self.add_documents(documents)

    def add_documents(self, documents: Iterable[Iterable[str]]):
        for doc in documents:
            uniq = set(doc)
            for tok in sorted(uniq):
                if tok not in self.token2id:
                    self.token2id[tok] = len(self.token2id)
            for tok in uniq:
                i = self.token2id[tok]
                self.dfs[i] = self.dfs.get(i, 0) + 1
            self.num_docs += 1

    def __len__(self):
        return len(self.token2id)


def read_documents_and_gen_idx_text(file_path: str) -> Tuple[List[List[str]], List[Tuple[List[str], List[int]]]]:
    """genmodel.py:21-43: split each line on ',', drop the path, keep documents with >= 3 tags, copy
    kept lines verbatim to <stem>_doc2vec_idx.csv; doc_id = running index of kept lines."""
    processed: List[List[str]] = []
    tagged: List[Tuple[List[str], List[int]]] = []
    idx_path = file_path.split(".")[0] + "_doc2vec_idx.csv"                 # :24
    with open(idx_path, "w", encoding="utf-8") as idx_f, open(file_path, "r", encoding="utf-8") as f:
        doc_id = 0
        for line in f:
            tokens = line.strip().split(",")[1:]                            # :29-31
            if tokens and len(tokens) >= 3:                                 # :36
                tagged.append((tokens, [doc_id]))
                processed.append(tokens)
                idx_f.write(line)
                idx_f.flush()
                doc_id += 1
    return processed, tagged


def read_documents(filename: str) -> List[str]:
    """genmodel.py:46-49."""
    with open(filename, "r", encoding="utf-8") as f:
        return [line.strip() for line in f.readlines()]


def count_non_empty_lines(file_path: str) -> int:
    """genmodel.py:101-107."""
    with open(file_path, "r", encoding="utf-8") as f:
        return sum(1 for line in f if line.strip())

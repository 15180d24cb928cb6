"""Helpers to read the golden fixtures written by oracle/make_fixtures.py."""
import os

import numpy as np
import torch

from oracle import fid_t5_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = z["dims"].tolist()
    dims = O.T5Dims(vocab_size=d[0], d_model=d[1], d_kv=d[2], d_ff=d[3], num_layers=d[4], num_decoder_layers=d[5],
                    num_heads=d[6], num_buckets=d[7], max_distance=d[8], dropout=0.0)
    w = {k[2:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith("w/")}
    return z, dims, w


def group(z, prefix):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


class FixTok:
    """deterministic stand-ins for nltk's WordPunctTokenizer / PorterStemmer (nltk is absent here and on the GPU box): the
    reference's stem_ems takes both as arguments, so its own code runs unchanged on them"""

    def tokenize(self, s):
        import re
        return re.findall(r"\w+|[^\w\s]+", s)


class FixStem:
    def stem(self, w):
        for suf in ("ing", "es", "s"):
            if w.endswith(suf) and len(w) - len(suf) >= 3:
                return w[:-len(suf)]
        return w

"""Tiny host helpers the choosers need (the reference pulls `log` from
spearmint/spearmint/helpers.py:10-14 together with protobuf plumbing that is
out of scope here)."""
from __future__ import absolute_import, print_function

import os
import sys
import tempfile

try:
    import cPickle as pickle  # Python 2
except ImportError:  # Python 3
    import pickle


def log(*args):
    """Write to stderr like helpers.py:10-14."""
    sys.stderr.write(" ".join(str(a) for a in args) + "\n")


def pickle_atomically(obj, path):
    """Dump to a temp file, then rename over `path` (the reference shells out
    to `mv` for NFS friendliness, GPEIChooser.py:66-83)."""
    d = os.path.dirname(os.path.abspath(path))
    fh = tempfile.NamedTemporaryFile(mode="wb", delete=False, dir=d)
    try:
        pickle.dump(obj, fh, protocol=2)
    finally:
        fh.close()
    os.rename(fh.name, path)


def unpickle(path):
    """Load a state pickle.  A file written by the Python-2 reference holds numpy arrays pickled as
    byte strings; Python 3 can only read those with encoding="latin1"."""
    with open(path, "rb") as fh:
        try:
            return pickle.load(fh)
        except UnicodeDecodeError:
            fh.seek(0)
            return pickle.load(fh, encoding="latin1")

"""Paths and constants (reference: src/config.py:1-25).  DATA_DIR / OUT_DIR can be
redirected with the VIBO_DATA_DIR / VIBO_OUT_DIR environment variables."""
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT_DIR = os.path.realpath(os.path.join(PKG_DIR, '..', '..'))
DATA_DIR = os.environ.get('VIBO_DATA_DIR', os.path.join(ROOT_DIR, 'data'))
OUT_DIR = os.environ.get('VIBO_OUT_DIR', os.path.join(ROOT_DIR, 'out'))
CHILDREN_LANG_DIR = os.path.join(DATA_DIR, 'critlangacq')

MISSING_DATA = -1   # responses use -1 for "missing"

# the reference omits '3pl_simulation' (KeyError at vibo.py:112 for --dataset 3pl_simulation); fixed here
IS_REAL_WORLD = {
    '1pl_simulation': False, '2pl_simulation': False, '3pl_simulation': False,
    'critlangacq': True, 'duolingo': True, 'wordbank': True, 'pisa2015_science': True,
    'score_matrix': True,       # (not in the reference: any pre-built [P, I] matrix, datasets.CachedScoreMatrix)
}

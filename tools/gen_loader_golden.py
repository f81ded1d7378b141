#!/usr/bin/env python3
"""Golden for vibo_amd.datasets.CachedScoreMatrix: the REAL reference loaders (src/datasets.py: DuoLingo_LanguageAcquisition,
WordBank_Language, PISAScience2015) run on a synthetic score-matrix CACHE -- the file each of them writes after parsing its
raw corpus and reads on every later run (datasets.py:505-515, 698-721, 751-817).  Build container only (imports /root/reference);
only inputs and the loaders' outputs are committed: tests/golden/score_matrix_loaders.npz.

Shims: stub `nltk` (imported, unused on the cached path); the *_DIR constants patched before src.datasets is imported;
WordBank reads its csv before it looks at the cache (datasets.py:649): a three-row csv with the columns it touches."""
import os
import sys
import tempfile
import types

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
REF = '/root/reference'


def main():
    rs = np.random.RandomState(7)
    P, I = 57, 23
    base = rs.randint(0, 2, size=(P, I)).astype(np.float64)
    base[rs.rand(P, I) < 0.3] = -1
    base[[3, 20, 41]] = -1                      # all-missing rows (dropped by the Duolingo / PISA loaders)
    duo = base.copy()
    obs = duo != -1
    duo[obs] = np.clip(duo[obs] * 0.6 + rs.rand(int(obs.sum())) * 0.4, 0, 1)      # Duolingo caches per-token averages: binarize rounds
    token_id = rs.permutation(1000)[:I]
    tmp = tempfile.mkdtemp()
    for sub in ('duolingo', 'wordbankr', 'pisa2015_science'):
        os.makedirs(os.path.join(tmp, sub))
    np.save(os.path.join(tmp, 'duolingo', 'score_matrix.npy'), duo)
    np.save(os.path.join(tmp, 'duolingo', 'token_id.npy'), token_id)
    np.save(os.path.join(tmp, 'wordbankr', 'score_matrix.npy'), base)
    np.save(os.path.join(tmp, 'pisa2015_science', 'score_matrix.npy'), base)
    with open(os.path.join(tmp, 'wordbankr', 'wordbankr_english.csv'), 'w') as f:
        f.write('data_id,num_item_id,value\n1,1,produces\n2,1,\n3,2,produces\n')
    nl = types.ModuleType('nltk')
    nl.word_tokenize = lambda s: s.split()
    sys.modules['nltk'] = nl
    sys.path.insert(0, REF)
    import src.config as cfg
    cfg.DATA_DIR = tmp
    cfg.DUOLINGO_LANG_DIR = os.path.join(tmp, 'duolingo')
    cfg.WORDBANKR_LANG_DIR = os.path.join(tmp, 'wordbankr')
    cfg.PISA2015_DIR = os.path.join(tmp, 'pisa2015_science')
    import src.datasets as D
    out = {'in.base': base, 'in.duolingo': duo, 'in.token_id': token_id}
    for name, cls in (('duolingo', D.DuoLingo_LanguageAcquisition), ('wordbank', D.WordBank_Language), ('pisa2015_science', D.PISAScience2015)):
        for tag, kw in (('train', dict(train=True)), ('test', dict(train=False)),
                        ('train_max', dict(train=True, max_num_person=30, max_num_item=11))):
            if name == 'wordbank' and 'max_num_item' in kw:
                kw = dict(train=True, max_num_person=30)      # (the reference's WordBank loader dies on max_num_item: `item_id` is unbound, datasets.py:686)
            ds = cls(**kw)
            out[f'{name}.{tag}.response'] = np.asarray(ds.response)
            out[f'{name}.{tag}.mask'] = np.asarray(ds.mask)
            idx, r, iid, m = ds[2]
            out[f'{name}.{tag}.item2.response'] = r.numpy()
            out[f'{name}.{tag}.item2.item_id'] = iid.numpy()
            out[f'{name}.{tag}.item2.mask'] = m.numpy()
            if hasattr(ds, 'item_id'):
                out[f'{name}.{tag}.item_id'] = np.asarray(ds.item_id)
    path = os.path.join(ROOT, 'tests', 'golden', 'score_matrix_loaders.npz')
    np.savez_compressed(path, **out)
    print(path, {k: v.shape for k, v in out.items() if k.endswith('.response') and 'item2' not in k})


if __name__ == '__main__':
    main()

import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from webrender_amd import scenes
from webrender_amd.harness import render_direct
from conftest import wrhip_lib, oracle_lib
kw = dict(n_lines=1, n_grads=1, n_lgrads=1, n_rgrads=0, n_cgrads=150, seed=171)
want,_ = render_direct(oracle_lib("gcc"), scenes.cache_decorations(**kw))
got,_ = render_direct(wrhip_lib(), scenes.cache_decorations(**kw))
d = np.abs(got['decoration_cache'].astype(int)-want['decoration_cache'].astype(int))
print('conic GPU vs oracle: max', d.max(), 'npx', int((d.max(axis=-1)>0).sum()))

import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, subprocess
if len(sys.argv) > 1:
    from webrender_amd import scenes, glapi
    from webrender_amd.harness import render_direct
    make={'images3': lambda: scenes.add_slivers(scenes.image_grid(seed=55), pitch=3),
          'gradients3': lambda: scenes.add_slivers(scenes.gradient_grid(), pitch=3, y1=1024)}[sys.argv[1]]
    got,st=render_direct(glapi.wrhip_path(),make())
    print(hex(st['gl_error']))
else:
    for scene in ('images3','gradients3'):
        for share in ('', '1'):
            out=[]
            for n in (100_000, 400_000, 1_600_000, 6_400_000, 25_600_000):
                env=dict(os.environ, WRHIP_RUNS_POOL_WORDS=str(n)); 
                if share: env['WRHIP_NO_RUN_SHARE']='1'
                r=subprocess.run([sys.executable, __file__, scene], env=env, capture_output=True, text=True)
                out.append((n, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else 'ERR'))
            print(scene, 'no_share' if share else 'share', out, flush=True)

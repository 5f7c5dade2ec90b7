import os, sys, subprocess
for env in ({}, {"DH_BOX_BAND": "64"}, {"DH_BOX_BAND": "32"}, {"DH_BOX_BAND": "96"}):
    e = dict(os.environ); e.update(env)
    for geo in (("320", "240", "1"), ("640", "480", "4")):
        out = subprocess.run([sys.executable, "tools/single_frame_sweep.py", *geo, "-:-"], env=e, capture_output=True, text=True).stdout
        import re
        for l in out.splitlines():
            if "traverse" in l: print(env, l)

"""The `sort` figure of `bin/bwtb3m verbose=1` at C3 against bench.py's ms_per_step: the CLI's one run is the FIRST launch of every kernel in its process
(code objects are loaded on first use), the bench's steps are warm.  python tools/cli_sort_time.py [nbases]"""
import os, subprocess, sys
sys.path.insert(0, "/root/repo")
from bwtb3m_amd import synth
nb = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
tmp = "/tmp/b3m_cli"; os.makedirs(tmp, exist_ok=True)
itype, d_text, n = synth.config_input_device("c3", size=nb, device="cuda")
pac = os.path.join(tmp, "c3.pac"); d_text.cpu().numpy().tofile(pac)
del d_text
import torch; torch.cuda.empty_cache()
for env in ({}, {"HIP_ENABLE_DEFERRED_LOADING": "0"}, {}):
    r = subprocess.run(["/root/repo/bin/bwtb3m", "inputtype=pacterm", "bwtonly=0", "numthreads=16", "verbose=1", "outputfilename=" + os.path.join(tmp, "o.bwt"), pac],
                       capture_output=True, text=True, env=dict(os.environ, **env))
    print(env, [l for l in r.stderr.splitlines() if "constructed" in l or "plan created" in l or "total" in l], flush=True)

"""k_lz stage time, fused against split form, for batches of N x 1 MiB (N from the command line): where the split form starts to pay."""
import os, subprocess, sys
for n in sys.argv[1:]:
    env = dict(os.environ, AB_N=n)
    out = subprocess.run([sys.executable, "scripts/ab.py", "PNA_LZ_SPLIT=0", "PNA_LZ_SPLIT=1"], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    print(n, " | ".join(l.split(" ", 1)[1] for l in out[-2:]), flush=True)

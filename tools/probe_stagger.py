"""Sweep the phase offset between the two co-resident workgroups of a CU (GLDM_R1D_STAGGER_US)."""
# Needs a diagnostic build of the library: `make -C graspldm_amd/csrc clean all EXTRA=-DGLDM_DEBUG_KNOBS`
# (the shipped build reads no environment variable: GLDM_R1D_STAGGER_US is compiled out).

import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for us in [int(v) for v in sys.argv[1:]] or [0, 50, 100, 150, 200, 250]:
    env = dict(os.environ, GLDM_R1D_STAGGER_US=str(us))
    out = subprocess.run([sys.executable, os.path.join(here, "run_denoise_once.py")], env=env, capture_output=True, text=True, timeout=120)
    print(f"stagger {us:4d} us: {out.stdout.strip()} {out.stderr.strip()[-200:]}")

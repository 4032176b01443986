"""Golden vectors of the reference's schedule helpers (utils.py:28-52), produced by RUNNING the reference's own utils.py (CPU).
Run in the build container only (needs /root/reference):   python tests/golden/make_golden_ramps.py
Writes ramps.npz (inputs + the reference's outputs; no source text)."""
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, REF)
    import utils as ref_utils
    cur = np.array([-3.0, 0.0, 0.5, 1.0, 2.5, 5.0, 7.75, 10.0, 12.0, 40.0])
    length = np.array([0.0, 1.0, 5.0, 10.0, 30.0])
    prog = np.array([-1.0, 0.0, 0.1, 0.25, 0.5, 0.625, 0.9, 1.0, 1.7])
    up = np.array([[ref_utils.sigmoid_rampup(c, L) for L in length] for c in cur])
    down = np.array([[ref_utils.cosine_rampdown(c, L) for L in length[1:]] for c in cur])
    rev = np.array([ref_utils.rev_sigmoid(p) for p in prog])
    sig = np.array([ref_utils.sigmoid(p) for p in prog])
    np.savez_compressed(os.path.join(OUT, "ramps.npz"), cur=cur, length=length, prog=prog, up=up, down=down, rev=rev, sig=sig)
    print("ramps.npz written", up.shape, down.shape)


if __name__ == "__main__":
    main()

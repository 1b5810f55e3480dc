"""The export wire format: `export-opencv` must reproduce, byte for byte, a
"vgg_generated_XX.i" header that the reference ships (workspace/opencv/vgg_generated_{48,64,80,120}.i) from
the inputs it was written from — the projection W of the reference's own result file and the
selected pooling-region filters (fixtures tests/golden/export_NN.npz, made by make_golden.py).
The pooling-region selection (src/misc.cpp:78-170) is exercised by hiding the 60 / 68 selected
filters among duplicates, all-zero filters and filters whose learned weight is not positive."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from util import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "opencv-dlco_amd", "cli", "export-opencv")


@pytest.fixture(scope="module")
def tool():
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(TOOL), "export-opencv"])
    return TOOL


def _write_inputs(tmp, z, rng):
    PR, W = z["PR"], z["W"]
    n_sel, cols = PR.shape
    gpos = (n_sel + 4 + 7) // 8                        # groups of 8 filter slots that carry a positive learned weight
    nw = gpos + 2                                      # ... and two groups whose weight is zero / negative
    filt = np.zeros((nw * 8, cols), np.float32)
    w = np.zeros((9, nw), np.float32)                  # the tool reads row `widx` of "w"
    slots = rng.permutation(gpos * 8)
    filt[slots[:n_sel]] = PR[rng.permutation(n_sel)]   # the selected filters, shuffled
    filt[slots[n_sel]] = PR[3]                         # duplicates of selected filters
    filt[slots[n_sel + 1]] = PR[41]
    # slots[n_sel+2:] stay all-zero filters
    filt[gpos * 8:] = rng.random((16, cols)).astype(np.float32)      # non-zero filters with weight <= 0
    widx = int(z["widx"])
    w[widx, :gpos] = rng.random(gpos).astype(np.float32) + 0.1
    w[widx, gpos], w[widx, gpos + 1] = 0.0, -0.5
    w[0, :] = 1.0                                      # another row of w must not matter
    side = int(round(np.sqrt(cols)))
    prg, prj = str(z["prg"]), str(z["prj"])
    for d in ("filters.h5", prg, prj):                 # directories of .npy files named like the reference's inputs
        os.makedirs(os.path.join(tmp, d), exist_ok=True)
    np.save(os.path.join(tmp, "filters.h5", "PRFilters.npy"), filt.reshape(nw * 8, side, side))
    np.save(os.path.join(tmp, prg, "w.npy"), w)
    np.save(os.path.join(tmp, prj, "W.npy"), W)
    return prg, widx, prj


@pytest.mark.parametrize("dim", [48, 64, 80, 120])
def test_export_reproduces_reference_header(tool, tmp_path, dim):
    """All four headers the reference ships (workspace/opencv/vgg_generated_{48,64,80,120}.i, workspace/11-opencv-export.sh)."""
    z = np.load(os.path.join(GOLDEN, "export_%d.npz" % dim))
    want = gzip.open(os.path.join(GOLDEN, "vgg_generated_%d.i.gz" % dim), "rb").read()
    prg, widx, prj = _write_inputs(str(tmp_path), z, np.random.default_rng(11 + dim))
    out = subprocess.run([tool, "-flt", "filters.h5", "-prg", prg, "-id", str(widx), "-prj", prj, "out.i"],
                         cwd=str(tmp_path), capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    n_sel, F = z["PR"].shape[0], z["W"].shape[1]
    assert ("PRFilters: %d x 4096 [%d]" % (n_sel, F)) in out.stdout and ("PJFilters: %d x [%d]" % (dim, F)) in out.stdout
    got = open(os.path.join(str(tmp_path), "out.i"), "rb").read()
    # All the headers the reference ships end with ONE more "\n" than src/export-opencv.cpp:372-388
    # writes (after the closing brace of PJ[] the source just closes the file).  This tool follows
    # the source; every byte before that final newline is identical, header comments included.
    assert want.endswith(b"\n};\n\n")
    assert got == want[:-1]


def test_export_argv_contract(tool, tmp_path):
    r = subprocess.run([tool, "-bogus"], capture_output=True, text=True)
    assert r.returncode == 1 and "ERROR: Invalid -bogus option." in r.stdout and "Usage: export-opencv -flt" in r.stdout
    r = subprocess.run([tool, "-flt", "a", "-prg", "b", "-prj", "c", "out.i"], capture_output=True, text=True)   # -id missing
    assert r.returncode == 1 and "Usage: export-opencv" in r.stdout


def test_export_dimension_mismatch(tool, tmp_path):
    z = dict(np.load(os.path.join(GOLDEN, "export_48.npz")))
    z["W"] = z["W"][:, :472]                           # W.cols != 8 * selected filters
    prg, widx, prj = _write_inputs(str(tmp_path), z, np.random.default_rng(12))
    r = subprocess.run([tool, "-flt", "filters.h5", "-prg", prg, "-id", str(widx), "-prj", prj, "out.i"],
                       cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0 and "ERROR: PJFilters [480] not agree PRFilters [472]." in r.stdout
    assert not os.path.exists(os.path.join(str(tmp_path), "out.i"))

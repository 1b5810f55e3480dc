"""The command-line surface (pj-learn, eval-fpr95) on a GPU: flags, exit codes, the stdout
grammar the reference's scripts scrape, and the W/A output files."""
import os
import re
import subprocess

import numpy as np
import pytest

from util import relmax, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "opencv-dlco_amd", "cli")

BEST = re.compile(r"^Best: (\d+)  Loss: (\d+\.\d{6}) Regul: (\d+\.\d{6}) Obj: (\d+\.\d{6}) \((\d+\.\d{6})\) Rank: (\d+) \((\d+)\) Ttime: \d+\.\d{4} Vtime: \d+\.\d{4}$")
STEP = re.compile(r"^Step: (\d+)  Loss: \d+\.\d{6} Regul: \d+\.\d{6} Obj: \d+\.\d{6} \(\d+\.\d{6}\) Rank: \d+ \(\d+\) Ttime: \d+\.\d{4} Vtime: \d+\.\d{4}$")
STAT = re.compile(r"^Stat: Dim \[(\d+)\] AUC: (\d\.\d{6}) \((\d\.\d{6})\) FPR95: (\d+\.\d{2}) \((\d+\.\d{2})\)( \[saved\])?$")


@pytest.fixture(scope="module")
def tools():
    subprocess.check_call(["make", "-s", "-C", CLI])
    return os.path.join(CLI, "pj-learn"), os.path.join(CLI, "eval-fpr95")


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    d = tmp_path_factory.mktemp("unproj")
    D, L = synth(4000, 64, k=10, seed=41, sp=0.7, noise=0.2)
    np.save(d / "Distance.npy", D)
    np.save(d / "Label.npy", L.reshape(-1, 1))
    return str(d), D, L


def test_usage_and_invalid_flag_exit_1(tools):
    pj, ev = tools
    for args in ([pj], [pj, "-bogus", "a", "b"], [pj, "only_one"], [pj, "-help", "a", "b"], [ev], [ev, "w_only"]):
        p = subprocess.run(args, capture_output=True, text=True)
        assert p.returncode == 1 and "Usage:" in p.stdout
    p = subprocess.run([pj, "-bogus", "a", "b"], capture_output=True, text=True)
    assert p.stdout.startswith("ERROR: Invalid -bogus option.")


def test_pj_learn_log_grammar_and_outputs(tools, dataset, tmp_path):
    pj, ev = tools
    src, D, L = dataset
    dst = str(tmp_path / "out")
    p = subprocess.run([pj, src, dst, "-mu", "0.004", "-gamma", "0.5", "-iters", "300", "-batch", "50"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()
    assert lines[0] == "mu: 0.004 gamma: 0.5 nIters: 300"
    assert lines[1] == "Load Labels: 4000" and lines[2] == "Load Distances: 4000 x 64"
    assert lines[3] == "0...10...20...30...40...50...60...70...80...90...100 - done."
    assert lines[4:10] == ["Positive samples #2000", "Negative samples #2000", "Positive train #1600",
                           "Negative train #1600", "Positive valid #400", "Negative valid #400"]
    assert lines[10] == "" and lines[11].startswith("Found GPU: ") and lines[12].startswith("Compute Capability: ") and lines[13] == ""
    body = lines[14:]
    steps = [l for l in body if l.startswith(("Best: ", "Step: "))]
    assert [int(re.split(r"[ :]+", l)[1]) for l in steps] == [100, 200, 300]           # LogStep = 100, t <= nIter
    saved = None
    for i, l in enumerate(body):
        if l.startswith("Best: "):
            assert BEST.match(l) and STAT.match(body[i + 1])
            if body[i + 1].endswith("[saved]"):
                saved = (BEST.match(l), STAT.match(body[i + 1]))
        elif l.startswith("Step: "):
            assert STEP.match(l)
        else:
            assert l.startswith("Stat: ")
    assert ": 300  Loss:" in p.stdout                    # the resume check of workspace/08-pjlearn.sh:17
    assert saved is not None
    W, A = np.load(dst + "/W.npy"), np.load(dst + "/A.npy")
    assert W.shape == (int(saved[1].group(1)), 64) and A.shape == (64, 64)
    assert relmax(W.T.astype(np.float64) @ W.astype(np.float64), A) <= 1e-5
    # Regul of the saved entry == mu * trace(A) (the reference's log <-> h5 consistency)
    assert abs(0.004 * np.trace(A.astype(np.float64)) - float(saved[0].group(3))) <= 2e-6

    # eval-fpr95 on the saved W reproduces the Stat line of the saved entry
    q = subprocess.run([ev, dst, src], capture_output=True, text=True, timeout=300)
    assert q.returncode == 0, q.stderr
    m = STAT.match(q.stdout.strip())
    assert m and int(m.group(1)) == W.shape[0]
    assert abs(float(m.group(2)) - float(saved[1].group(2))) <= 2e-6 and abs(float(m.group(4)) - float(saved[1].group(4))) <= 0.011


def test_hdf5_round_trip_when_libhdf5_is_present(tools, dataset, tmp_path):
    import ctypes.util
    if not (os.path.exists("/opt/conda/lib/libhdf5.so") or ctypes.util.find_library("hdf5")):
        pytest.skip("no libhdf5 on this box")
    pj, ev = tools
    src, _, _ = dataset
    dst = str(tmp_path / "out.h5")
    p = subprocess.run([pj, src, dst, "-mu", "0.004", "-iters", "100", "-batch", "50"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    assert os.path.getsize(dst) > 64 * 64 * 4
    q = subprocess.run([ev, dst, src], capture_output=True, text=True, timeout=300)
    assert q.returncode == 0 and STAT.match(q.stdout.strip()), q.stderr

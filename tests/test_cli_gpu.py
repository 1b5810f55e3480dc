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


def _last_entry(stdout):
    best = [l for l in stdout.splitlines() if l.startswith(("Best: ", "Step: "))]
    f = re.split(r"[ :()]+", best[-1])
    # Best 100 Loss x Regul y Obj z (zb) Rank r (rb) ...
    return dict(t=int(f[1]), loss=float(f[3]), regul=float(f[5]), rank=int(f[10]))


def test_pj_learn_two_processes_on_one_gpu_match_the_single_process_run(tools, dataset, tmp_path):
    """`pj-learn -gpus 2`: the parent forks two ranks before any GPU call; each runs the column-sharded
    step on its own context and the all-gathers go through the library.  A box of this pool has one
    GPU and RCCL refuses two ranks on one device, so the ranks share device 0 and use the library's
    shared-memory transport (-comm host); with -comm rccl the same processes call ncclAllGather.
    Same global batch as the single-process run => the same computation up to fp32 summation grouping."""
    pj, _ = tools
    src, D, L = dataset
    args = ["-mu", "0.004", "-gamma", "0.5", "-iters", "100", "-batch", "40"]
    one = subprocess.run([pj, src, str(tmp_path / "one")] + args, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr
    two = subprocess.run([pj, src, str(tmp_path / "two"), "-gpus", "2", "-devices", "0,0", "-comm", "host"] + args,
                         capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr
    # one log, printed by rank 0 only, same grammar and header
    assert two.stdout.splitlines()[:13] == one.stdout.splitlines()[:13]
    a, b = _last_entry(one.stdout), _last_entry(two.stdout)
    assert a["t"] == b["t"] == 100 and abs(a["rank"] - b["rank"]) <= 1
    assert abs(a["loss"] - b["loss"]) <= 0.02 * a["loss"] + 1e-5 and abs(a["regul"] - b["regul"]) <= 0.02 * a["regul"] + 1e-5
    W1, W2 = np.load(tmp_path / "one" / "W.npy"), np.load(tmp_path / "two" / "W.npy")
    A1, A2 = np.load(tmp_path / "one" / "A.npy"), np.load(tmp_path / "two" / "A.npy")
    assert abs(W1.shape[0] - W2.shape[0]) <= 1 and A1.shape == A2.shape
    assert relmax(A2, A1) <= 2e-2                      # 100 free-running steps on either side
    assert relmax(W2.T.astype(np.float64) @ W2.astype(np.float64), A2) <= 1e-5


def test_pj_learn_two_processes_allreduce_layout_matches_the_single_process_run(tools, dataset, tmp_path):
    """`pj-learn -gpus 2 -dp allreduce`: the exchange BASELINE configs[3] words, from C++ - every rank keeps the whole dual
    average, the library all-gathers the 2B distances and sum-all-reduces the F x F partial gradients each step
    (ncclAllReduce with -comm rccl; here, two ranks on the one GPU of the box, the shared-memory transport, which adds the
    ranks' pieces in rank order).  Same global batch as the single process => same computation up to summation grouping."""
    pj, _ = tools
    src, D, L = dataset
    args = ["-mu", "0.004", "-gamma", "0.5", "-iters", "100", "-batch", "40"]
    one = subprocess.run([pj, src, str(tmp_path / "one")] + args, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr
    two = subprocess.run([pj, src, str(tmp_path / "two"), "-gpus", "2", "-devices", "0,0", "-comm", "host", "-dp", "allreduce"] + args,
                         capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr
    assert two.stdout.splitlines()[:13] == one.stdout.splitlines()[:13]
    a, b = _last_entry(one.stdout), _last_entry(two.stdout)
    assert a["t"] == b["t"] == 100 and abs(a["rank"] - b["rank"]) <= 1
    assert abs(a["loss"] - b["loss"]) <= 0.02 * a["loss"] + 1e-5 and abs(a["regul"] - b["regul"]) <= 0.02 * a["regul"] + 1e-5
    A1, A2 = np.load(tmp_path / "one" / "A.npy"), np.load(tmp_path / "two" / "A.npy")
    W2 = np.load(tmp_path / "two" / "W.npy")
    assert relmax(A2, A1) <= 2e-2                      # 100 free-running steps on either side
    assert relmax(W2.T.astype(np.float64) @ W2.astype(np.float64), A2) <= 1e-5
    bad = subprocess.run([pj, src, str(tmp_path / "x"), "-gpus", "2", "-dp", "bogus"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "Usage:" in bad.stdout


def test_pj_learn_multi_process_failure_exits_nonzero(tools, dataset, tmp_path):
    """A rank that cannot start (device 99 does not exist) exits non-zero; the parent stops the other
    rank instead of leaving it in a collective, and reports failure."""
    pj, _ = tools
    src, _, _ = dataset
    p = subprocess.run([pj, src, str(tmp_path / "bad"), "-gpus", "2", "-devices", "0,99", "-comm", "host", "-iters", "100", "-batch", "40"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 3
    assert "rank 1" in p.stderr
    q = subprocess.run([pj, src, str(tmp_path / "bad"), "-gpus", "2", "-batch", "41"], capture_output=True, text=True, timeout=60)
    assert q.returncode == 1 and "Usage:" in q.stdout


def test_pj_learn_rccl_rank_failure_does_not_kill_the_parent(tools, dataset, tmp_path):
    """-comm rccl: rank 1 (device 99) dies before it reads the ncclUniqueId, so the parent's relay writes into a
    pipe without a reader.  The parent ignores SIGPIPE, sees the failed rank, stops rank 0 (which would sit in
    ncclCommInitRank for ever) and exits with 3 - not by signal - leaving no child behind."""
    import ctypes.util
    if not (ctypes.util.find_library("rccl") or os.path.exists("/opt/rocm/lib/librccl.so")):
        pytest.skip("no librccl on this box")
    pj, _ = tools
    src, _, _ = dataset
    p = subprocess.run([pj, src, str(tmp_path / "bad"), "-gpus", "2", "-devices", "0,99", "-comm", "rccl", "-iters", "100", "-batch", "40"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 3, (p.returncode, p.stderr[-400:])
    assert "rank 1" in p.stderr
    left = subprocess.run(["pgrep", "-x", "pj-learn"], capture_output=True, text=True)
    assert left.stdout.strip() == "", "a rank survived its parent: " + left.stdout


def test_pr_learn_logs_at_every_multiple_of_logstep_including_one(tmp_path):
    """src/pr-learn.cpp:331,419-422: the step counter restarts at 0 inside the logging iteration and is incremented in
    the same iteration, so the lines come at t = LogStep, 2 LogStep, ... - with -logstep 1 at EVERY t >= 1."""
    subprocess.check_call(["make", "-s", "-C", CLI])
    pr = os.path.join(CLI, "pr-learn")
    N, F = 400, 64
    rng = np.random.default_rng(9)
    L = (np.arange(N) % 2 == 0).astype(np.uint8)
    D = rng.random((N, F)).astype(np.float32)
    D[:, :8] *= np.where(L[:, None] == 1, 0.25, 1.0).astype(np.float32)
    flt, src = tmp_path / "filters", tmp_path / "fulldists"
    flt.mkdir(); src.mkdir()
    np.save(flt / "PRParams.npy", rng.integers(0, 6, (8 * F, 3)).astype(np.float32))
    np.save(flt / "RingParams.npy", np.zeros((5, 3), np.float32))
    np.save(src / "Distance.npy", D)
    np.save(src / "Label.npy", L.reshape(-1, 1))
    for logstep, iters, want in ((1, 6, [1, 2, 3, 4, 5, 6]), (3, 10, [3, 6, 9]), (4, 3, [])):
        q = subprocess.run([pr, str(flt), str(src), str(tmp_path / ("o%d" % logstep)), "-iters", str(iters), "-logstep", str(logstep), "-maxdim", "100000"],
                           capture_output=True, text=True, timeout=300)
        assert q.returncode == 0, q.stderr
        ts = [int(re.split(r"[ :]+", l)[1]) for l in q.stdout.splitlines() if l.startswith(("Best: ", "Step: "))]
        assert ts == want, (logstep, ts)
    z = subprocess.run([pr, str(flt), str(src), str(tmp_path / "o0"), "-logstep", "0"], capture_output=True, text=True, timeout=60)
    assert z.returncode == 1 and "Usage:" in z.stdout


def test_pr_learn_cli_grammar_and_saved_rows(tmp_path):
    """pr-learn: flags, usage, the Best/Stat/Step grammar of src/pr-learn.cpp:366-403 (scraped by
    workspace/05-prstats.sh) and the rows appended to the "w" dataset, one per "[saved]" line."""
    subprocess.check_call(["make", "-s", "-C", CLI])
    pr = os.path.join(CLI, "pr-learn")
    p = subprocess.run([pr, "only", "two"], capture_output=True, text=True)
    assert p.returncode == 1 and "Usage: pr-learn  src_h5_filter_file" in p.stdout
    N, F = 3000, 256
    rng = np.random.default_rng(4)
    L = (np.arange(N) % 2 == 0).astype(np.uint8)
    D = rng.random((N, F)).astype(np.float32)
    D[:, :24] *= np.where(L[:, None] == 1, 0.25, 1.0).astype(np.float32)
    P = rng.integers(0, 6, (8 * F, 3)).astype(np.float32)
    flt, src = tmp_path / "filters", tmp_path / "fulldists"
    flt.mkdir(); src.mkdir()
    np.save(flt / "PRParams.npy", P)
    np.save(flt / "RingParams.npy", np.zeros((5, 3), np.float32))
    np.save(src / "Distance.npy", D)
    np.save(src / "Label.npy", L.reshape(-1, 1))
    for dst in (str(tmp_path / "out"), str(tmp_path / "out.h5")):
        if dst.endswith(".h5") and not os.path.exists("/opt/conda/lib/libhdf5.so"):
            continue
        q = subprocess.run([pr, str(flt), str(src), dst, "-mu", "0.03", "-gamma", "0.25", "-iters", "20000", "-logstep", "4000", "-maxdim", "100000"],
                           capture_output=True, text=True, timeout=600)
        assert q.returncode == 0, q.stderr
        lines = q.stdout.splitlines()
        assert lines[0] == "mu: 0.03 gamma: 0.25 maxdim: 100000 nIters: 20000"
        assert lines[1] == "Load PRParams." and lines[2] == "Load RingParams."
        assert lines[3] == "Load Labels: 3000" and lines[4] == "Load Distances: 3000 x 256"
        body = [l for l in lines if l.startswith(("Best: ", "Step: ", "Stat: "))]
        ts = [int(re.split(r"[ :]+", l)[1]) for l in body if l.startswith(("Best: ", "Step: "))]
        assert ts == [4000, 8000, 12000, 16000, 20000]
        best = re.compile(r"^Best: \d+  Loss: \d+\.\d{6} Regul: \d+\.\d{6} Obj: \d+\.\d{6} \(\d+\.\d{6}\)  NNZ: \d+ \(\d+\)  Ttime: \d+\.\d{4} Vtime: \d+\.\d{4}$")
        stat = re.compile(r"^Stat: nPR #\d+ \(#\d+\) Dim/MaxDim \[\d+/100000\] AUC: \d\.\d{6} FPR95: \d+\.\d{2}( \[saved\])?$")
        step = re.compile(r"^Step: \d+  Loss: \d+\.\d{6} Regul: \d+\.\d{6} Obj: \d+\.\d{6} \(\d+\.\d{6}\)  NNZ: \d+ \(\d+\)  Ttime: \d+\.\d{4} Vtime: \d+\.\d{4}$")
        for i, l in enumerate(body):
            assert best.match(l) or stat.match(l) or step.match(l), l
            if l.startswith("Best: "):
                assert stat.match(body[i + 1])
        nsaved = sum(l.endswith("[saved]") for l in body)
        if not dst.endswith(".h5"):
            assert nsaved >= 1
            W = np.load(dst + "/w.npy")
            assert W.shape == (nsaved, F) and (W >= 0).all() and (W != 0).any()
        else:
            assert os.path.getsize(dst) > 0


def test_comp_uprjdists_cli_feeds_pj_learn(tmp_path):
    """comp-uprjdists: the reference's flags and dataset names (src/comp-uprjdists.cpp:54-133,254-349); its
    "Distance"/"Label" equal the library's descriptors differenced per pair, and pj-learn trains on them."""
    import importlib
    from test_descriptors import make_filters, make_patches
    dlco = importlib.import_module("opencv-dlco_amd")
    subprocess.check_call(["make", "-s", "-C", CLI])
    cu, pj = os.path.join(CLI, "comp-uprjdists"), os.path.join(CLI, "pj-learn")
    p = subprocess.run([cu, "only"], capture_output=True, text=True)
    assert p.returncode == 1 and "Usage: comp-uprjdists src_h5_filter_file src_h5_img_file" in p.stdout
    n, wcols = 120, 6
    rng = np.random.default_rng(12)
    patches = make_patches(n, seed=31)
    PR = np.zeros((wcols * 8, 4096), np.float32)
    PR[:40] = make_filters(40, seed=3, scale=25.0)
    PR[5] = PR[4]                                                  # a repeat and all-zero rows (40..47)
    w = np.array([[0.3, 0.0, 0.2, 0.1, 0.5, 0.7], [0, 0, 0, 0, 1.0, 0], [0, 0, 0, 0, 0, 1.0]], np.float32)
    ids = rng.integers(0, 12, n)
    a, b = rng.integers(0, n, 900), rng.integers(0, n, 900)
    b[::2] = [rng.choice(np.nonzero(ids == ids[x])[0]) for x in a[::2]]
    pairs = np.stack([a, ids[a], b, ids[b]], 1).astype(np.int32)
    flt, img, prj, out = (tmp_path / s for s in ("filters", "images", "prj", "dists"))
    for d in (flt, img, prj):
        d.mkdir()
    np.save(flt / "PRFilters.npy", PR.reshape(-1, 64, 64))
    np.save(img / "Indices.npy", pairs)
    np.save(img / "Patches.npy", patches)
    np.save(prj / "w.npy", w)
    q = subprocess.run([cu, str(flt), str(img), "-prj", str(prj), "-id", "0", "-out", str(out)], capture_output=True, text=True, timeout=600)
    assert q.returncode == 0, q.stderr
    sPR = dlco.select_filters(PR, w[0])
    assert len(sPR) == 31                                          # 4 columns x 8 rows, minus the repeat
    assert "PRFilters: 31 x 4096" in q.stdout and "Descriptor size: 248" in q.stdout and "Done." in q.stdout
    D, L = np.load(out / "Distance.npy"), np.load(out / "Label.npy")
    ctx = dlco.DescContext()
    ctx.set_filters(sPR)
    want, lab = ctx.pair_dists(patches, pairs)
    assert D.shape == (900, 248) and np.array_equal(D, want)
    assert L.shape == (900, 1) and np.array_equal(L.ravel(), lab) and 0 < L.sum() < 900
    # -id picks the row of "w"
    q = subprocess.run([cu, str(flt), str(img), "-prj", str(prj), "-id", "1", "-out", str(tmp_path / "d1")], capture_output=True, text=True, timeout=600)
    assert q.returncode == 0 and "Descriptor size: 64" in q.stdout
    # row 2 of w selects only all-zero filters, row 3 does not exist
    q = subprocess.run([cu, str(flt), str(img), "-prj", str(prj), "-id", "2", "-out", str(tmp_path / "d2")], capture_output=True, text=True)
    assert q.returncode == 2 and "no pooling region selected" in q.stderr
    q = subprocess.run([cu, str(flt), str(img), "-prj", str(prj), "-id", "3", "-out", str(tmp_path / "d3")], capture_output=True, text=True)
    assert q.returncode == 2 and "no such row" in q.stderr
    # the next stage of the pipeline reads what this one wrote
    r = subprocess.run([pj, str(out), str(tmp_path / "model"), "-iters", "40", "-batch", "32"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "Load Distances: 900 x 248" in r.stdout


def test_full_pipeline_of_the_reference_tools(tmp_path):
    """BASELINE configs[4] as plumbing: comp-fulldists -> pr-learn -> comp-uprjdists -> pj-learn -> eval-fpr95,
    every stage reading the files the previous one wrote, with the dataset names of the reference's
    workspace scripts; the learned projection is scored on a SECOND patch set (cross-set evaluation)."""
    from test_descriptors import make_filters, make_patches
    subprocess.check_call(["make", "-s", "-C", CLI])
    tool = lambda t: os.path.join(CLI, t)
    regions = 16

    def patch_set(seed, n=160, npairs=1200):
        rng = np.random.default_rng(seed)
        base = make_patches(40, seed=seed)                              # 40 "3D points", 4 noisy views each
        patches = np.concatenate([np.clip(base.astype(np.float32) + rng.normal(0, 4 + 3 * v, base.shape), 0, 255).astype(np.uint8)
                                  for v in range(n // 40)])
        ids = np.tile(np.arange(40), n // 40)
        a, b = rng.integers(0, n, npairs), rng.integers(0, n, npairs)
        b[::2] = [rng.choice(np.nonzero(ids == ids[x])[0]) for x in a[::2]]
        return patches, np.stack([a, ids[a], b, ids[b]], 1).astype(np.int32)

    PR = make_filters(8 * regions, seed=17, scale=20.0)
    PRParams = np.zeros((8 * 8 * regions, 3), np.float32)
    PRParams[:, 0] = 1 + (np.arange(8 * 8 * regions) % 3)
    flt = tmp_path / "filters"
    flt.mkdir()
    np.save(flt / "PRFilters.npy", PR.reshape(-1, 64, 64))
    np.save(flt / "PRParams.npy", PRParams)
    np.save(flt / "RingParams.npy", np.zeros((5, 3), np.float32))
    sets = {}
    for name, seed in (("train", 1), ("test", 2)):
        d = tmp_path / name
        d.mkdir()
        patches, pairs = patch_set(seed)
        np.save(d / "Patches.npy", patches)
        np.save(d / "Indices.npy", pairs)
        sets[name] = d
    run = lambda args: subprocess.run(args, capture_output=True, text=True, timeout=900)
    from oracle import ref
    tr_patches, tr_pairs = np.load(sets["train"] / "Patches.npy"), np.load(sets["train"] / "Indices.npy")
    # 1. pooling-region distances of the training set == the oracle's restatement of comp-fulldists, pair by pair
    r = run([tool("comp-fulldists"), str(flt), str(sets["train"]), str(tmp_path / "fulldists")])
    assert r.returncode == 0 and "Bins: #8 Sigma: 1.4 bNorm: 1" in r.stdout, r.stderr
    full = np.load(tmp_path / "fulldists" / "Distance.npy")
    flab = np.load(tmp_path / "fulldists" / "Label.npy").ravel()
    assert full.shape == (1200, regions) and (full >= 0).all()
    sub = np.arange(0, 1200, 5)
    want_full = np.stack([ref.full_dists(tr_patches[a], tr_patches[b], PR) for a, _, b, _ in tr_pairs[sub]])
    assert np.abs(full[sub] - want_full).max() <= 1e-5 * want_full.max()           # cuda::gemm / reduce order unspecified there
    assert np.array_equal(flab, (tr_pairs[:, 1] == tr_pairs[:, 3]).astype(np.uint8))
    # 2. pooling-region selection: every row pr-learn saved is, bit for bit, the oracle's w of that iteration, and the
    #    log's Loss / Regul / NNZ are the oracle's validation numbers for it (src/pr-learn.cpp:302-369)
    r = run([tool("pr-learn"), str(flt), str(tmp_path / "fulldists"), str(tmp_path / "prs"), "-mu", "0.001", "-gamma", "0.5", "-iters", "6000",
             "-logstep", "2000", "-maxdim", "100000"])
    assert r.returncode == 0 and "[saved]" in r.stdout, r.stdout + r.stderr
    w = np.load(tmp_path / "prs" / "w.npy")
    assert w.shape[1] == regions and (w[-1] > 0).any()
    prt = ref.PrTrainer(full, flab, mu=0.001, gamma=0.5)
    lines = r.stdout.splitlines()
    k = done = 0
    for i, l in enumerate(lines):
        m = re.match(r"^(Best|Step): (\d+)  Loss: (\S+) Regul: (\S+) Obj: \S+ \(\S+\)  NNZ: (\d+) ", l)
        if not m:
            continue
        t = int(m.group(2))
        prt.steps(t + 1 - done)                                        # iterations 0..t have run when the line is printed
        done = t + 1
        lo, rg, nz = prt.validate()
        assert abs(lo - float(m.group(3))) <= 1.5e-6 and abs(rg - float(m.group(4))) <= 1.5e-6 and nz == int(m.group(5)), l
        if m.group(1) == "Best" and lines[i + 1].endswith("[saved]"):
            assert np.array_equal(w[k], prt.state()["w"]), "saved row %d differs from the oracle's w at t = %d" % (k, t)
            k += 1
    assert k == len(w) >= 1
    prt.close()
    # 3. descriptors of the selected regions, both sets == the oracle's get_desc + pooling + clamp, differenced per pair
    for name in sets:
        r = run([tool("comp-uprjdists"), str(flt), str(sets[name]), "-prj", str(tmp_path / "prs"), "-id", str(len(w) - 1), "-out", str(tmp_path / (name + "-unproj"))])
        assert r.returncode == 0, r.stderr
    D = np.load(tmp_path / "train-unproj" / "Distance.npy")
    Ltr = np.load(tmp_path / "train-unproj" / "Label.npy").ravel()
    assert D.shape[0] == 1200 and D.shape[1] % 8 == 0
    sPR = ref.select_pr_filters(PR, w[-1])
    assert D.shape[1] == 8 * len(sPR)
    sub = np.arange(0, 1200, 7)
    descs = {}
    for a, _, b, _ in tr_pairs[sub]:
        for pid in (a, b):
            if pid not in descs:
                descs[pid] = ref.patch_descriptor(tr_patches[pid], sPR)
    want_D = np.stack([descs[a] - descs[b] for a, _, b, _ in tr_pairs[sub]])
    assert np.abs(D[sub] - want_D).max() <= 6e-7                       # descriptors are <= 1: one float rounding of either side
    assert np.array_equal(Ltr, (tr_pairs[:, 1] == tr_pairs[:, 3]).astype(np.uint8))
    # 4. the projection: pj-learn's saved model against the oracle's free-running trainer on the same file (the hinge mask
    #    makes trajectories chaotic: rank, objective and FPR@95 in the band the sample size allows), and its own consistency
    r = run([tool("pj-learn"), str(tmp_path / "train-unproj"), str(tmp_path / "model"), "-iters", "150", "-batch", "64", "-logstep", "50", "-mu", "0.002"])
    assert r.returncode == 0 and "[saved]" in r.stdout, r.stdout + r.stderr
    Wm, Am = np.load(tmp_path / "model" / "W.npy"), np.load(tmp_path / "model" / "A.npy")
    assert relmax(Wm.T.astype(np.float64) @ Wm.astype(np.float64), Am) <= 1e-5
    last = _last_entry(r.stdout)
    otr = ref.Trainer(D, Ltr, B=64, mu=0.002, gamma=0.5, grad_order=1)
    for _ in range(151):                                               # t = 0 .. 150
        otr.step()
    lo_o, rg_o = otr.validate()
    dim_o, f95_o, auc_o = otr.stats()
    assert last["t"] == 150 and abs(last["rank"] - dim_o) <= max(3, dim_o // 10), (last, dim_o)
    assert abs(last["loss"] - lo_o) <= 0.15 * lo_o + 1e-3 and abs(last["regul"] - rg_o) <= 0.15 * rg_o + 1e-4, (last, lo_o, rg_o)
    otr.close()
    # 5. evaluation: eval-fpr95's line for the saved W is the oracle's ComputePJStats of that W, on the training set and
    #    on the SECOND patch set (cross-set), over the distances the oracle itself computes
    for name in ("train", "test"):
        ev = run([tool("eval-fpr95"), str(tmp_path / "model"), str(tmp_path / (name + "-unproj"))])
        assert ev.returncode == 0, ev.stderr
        m = STAT.match(ev.stdout.strip())
        assert m, ev.stdout
        De = np.load(tmp_path / (name + "-unproj") / "Distance.npy")
        Le = np.load(tmp_path / (name + "-unproj") / "Label.npy").ravel()
        d_o = ref.project_sqdist(Wm, De)
        f_o, a_o = ref.roc_stats(d_o, Le)
        assert int(m.group(1)) == Wm.shape[0]
        # FPR95 is printed in per cent with two decimals; ties in the ranking may move one row
        assert abs(float(m.group(4)) - 100.0 * f_o) <= 0.006 + 100.0 / max(int((Le == 0).sum()), 1), (name, m.group(4), f_o)
        assert abs(float(m.group(2)) - a_o) <= 1e-4, (name, m.group(2), a_o)
        print("pipeline: %s set FPR95 %s %% (oracle %.2f %%), AUC %s (oracle %.6f)" % (name, m.group(4), 100 * f_o, m.group(2), a_o))

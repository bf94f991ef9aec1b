"""Snapshot text files of the host layer (wavenet_amd/host/src/Snapshot.cxx; reference
src/Snapshot.cxx:54-158, src/Wavenet.cxx:136-178).  No GPU is needed: saving and loading touch host
state only.

tests/golden/reference_layout.snap is hand-written in the layout the reference's writer produces --
scalars through the default stream format, vectors one right-aligned number per line with 4 decimal
digits (fixed or scientific), each closed by an empty line and "#".  It was not produced by the
reference (its Armadillo dependency is absent here); what it pins is the reader, which follows the
reference's token-by-token algorithm.
"""
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "cpp", "host_driver")
FIXTURE = os.path.join(ROOT, "tests", "golden", "reference_layout.snap")


def reference_reader(path):
    """The reference's operator>> (src/Snapshot.cxx:83-158), token for token."""
    tokens = open(path).read().split()
    pos = 0

    def number(tok):
        try:
            return float(tok)
        except ValueError:
            return None

    out = {}
    out["scalars"] = [float(t) for t in tokens[:4]]
    pos = 4
    for name in ("filter", "momentum"):
        vec = []
        while pos < len(tokens) and "#" not in tokens[pos]:
            if number(tokens[pos]) is not None:
                vec.append(number(tokens[pos]))
            pos += 1
        pos += 1
        out[name] = vec
    out["batchSize"] = int(tokens[pos])
    pos += 1
    tmp = tokens[pos]
    pos += 1
    for name, stop in (("batchQueue", "FILTERLOG"), ("filterLog", "COSTLOG")):
        out[name] = []
        while stop not in tmp:
            vec = []
            while pos < len(tokens):
                tmp = tokens[pos]
                pos += 1
                if number(tmp) is None:
                    break
                vec.append(number(tmp))
            if not vec:
                break
            out[name].append(vec)
    out["costLog"] = [float(t) for t in tokens[pos:]]
    return out


def run_snapshot(infile, pattern, number, cwd):
    assert os.path.exists(DRIVER), "build it with make -C tests/cpp (done by __graft_entry__.build())"
    res = subprocess.run([DRIVER, "snapshot", infile, pattern, str(number)], capture_output=True, text=True,
                         timeout=120, cwd=cwd)
    assert res.returncode == 0, (res.returncode, res.stdout[-500:], res.stderr[-2000:])
    out = {}
    for line in res.stdout.splitlines():
        parts = line.split()
        if parts[0] == "file":
            out["file"] = parts[1]
            continue
        out[parts[0]] = np.array([float(v) for v in parts[2:]])
        assert len(out[parts[0]]) == int(parts[1])
    return out, res.stderr


def test_reads_the_reference_layout_and_round_trips():
    want = reference_reader(FIXTURE)
    assert want["batchSize"] == 4 and len(want["batchQueue"]) == 2 and len(want["filterLog"]) == 2
    with tempfile.TemporaryDirectory() as tmp:
        out, err = run_snapshot(FIXTURE, "run/snapshots/test.%06u.snap", 7, tmp)
        # loaded state = what the reference's reader takes from the same file
        assert np.array_equal(out["filter"], want["filter"])
        assert np.array_equal(out["momentum"], want["momentum"])
        assert np.array_equal(out["misc"], [want["scalars"][0], want["scalars"][1], want["batchSize"]])
        assert np.array_equal(out["filterLog"], np.concatenate(want["filterLog"]))
        assert np.array_equal(out["costLog"], want["costLog"])
        # file naming: the pattern completed with the number, post-increment steps to the next one
        assert out["file"] == "run/snapshots/test.000007.snap"
        assert list(out["exists"]) == [0, 1] and out["nextNumber"][0] == 8
        assert "does not exist. Creating it." in err
        # the written file, read by the reference's algorithm, holds the same state in full precision
        got = reference_reader(os.path.join(tmp, out["file"]))
        assert got == want
        text = open(os.path.join(tmp, out["file"])).read()
        assert text.count("#") == 2 + 2 + 2 and "BATCHQUEUE\n" in text and "FILTERLOG\n" in text and "COSTLOG\n" in text
        # and loads back into a second object unchanged
        for key in ("filter", "momentum", "misc", "filterLog", "costLog"):
            assert np.array_equal(out["re_" + key], out[key])
        # a missing file leaves the object alone; absolute paths are refused (src/Wavenet.cxx:141-144, 169-172)
        assert list(out["missingKept"]) == [1.0, 2.0]
        assert out["absoluteWritten"][0] == 0
        assert "doesn't exists" in err and "Only accepting realtive paths" in err


def test_full_precision_survives():
    """Numbers that the reference's 4-digit vector format would truncate come back bit for bit."""
    rng = np.random.default_rng(3)
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "in.snap")
        filt = rng.standard_normal(6) * 1e-3
        mom = rng.standard_normal(6) * 1e5
        queue = [rng.standard_normal(6) for _ in range(3)]
        flog = [rng.standard_normal(6) for _ in range(5)]
        clog = list(rng.random(4)) + [0.0]
        with open(src, "w") as f:
            f.write("%r\n%r\n%r\n%r\n" % (12.5, 1 / 3, 0.1, 7.0))
            for vec in (filt, mom):
                f.write("".join("%r\n" % float(v) for v in vec) + "\n#\n")
            f.write("16\nBATCHQUEUE\n")
            for vec in queue:
                f.write("".join("%r\n" % float(v) for v in vec) + "\n#\n")
            f.write("FILTERLOG\n")
            for vec in flog:
                f.write("".join("%r\n" % float(v) for v in vec) + "\n#\n")
            f.write("COSTLOG\n" + "".join("%r\n" % float(v) for v in clog))
        out, _ = run_snapshot(src, "plain.snap", 0, tmp)
        assert out["file"] == "plain.snap"
        assert np.array_equal(out["filter"], filt) and np.array_equal(out["momentum"], mom)
        assert np.array_equal(out["filterLog"], np.concatenate(flog)) and np.array_equal(out["costLog"], clog)
        assert np.array_equal(out["misc"], [12.5, 1 / 3, 16])
        got = reference_reader(os.path.join(tmp, "plain.snap"))
        assert got == reference_reader(src)
        assert [list(q) for q in queue] == got["batchQueue"]


def test_empty_queue_and_logs():
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "in.snap")
        with open(src, "w") as f:
            f.write("10\n0.001\n0\n0\n   0.7071\n   0.7071\n\n#\n        0\n        0\n\n#\n1\nBATCHQUEUE\nFILTERLOG\n"
                    "   0.7071\n   0.7071\n\n#\nCOSTLOG\n0\n")
        out, _ = run_snapshot(src, "again.snap", 0, tmp)
        assert list(out["filter"]) == [0.7071, 0.7071] and list(out["momentum"]) == [0, 0]
        assert list(out["filterLog"]) == [0.7071, 0.7071] and list(out["costLog"]) == [0]
        got = reference_reader(os.path.join(tmp, "again.snap"))
        assert got["batchQueue"] == [] and got["filterLog"] == [[0.7071, 0.7071]] and got["costLog"] == [0]

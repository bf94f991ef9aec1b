"""Properties of the compiled gfx950 code object that results depend on, read from the built library without a
GPU: the register-heavy kernels of the streamed ranking must keep everything in registers (no private
segment).  kr_bucket returned wrong sums, differing from run to run, in the builds where the compiler had
spilled a few registers to scratch memory at 6 waves per SIMD; a change that brings the spills back should
fail here, not as a flaky parity test on the GPU."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

LIB = os.path.join(os.path.dirname(__file__), "..", "wavenet_amd", "lib", "libwavenet_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_table():
    """{mangled kernel name: (private segment bytes, VGPRs)} over every gfx950 code object bundled in the library."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        lib = shutil.copy(LIB, tmp)                      # the extractor writes next to its input
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", lib], check=True, capture_output=True, cwd=tmp)
        for name in sorted(os.listdir(tmp)):
            if "amdgcn" not in name:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(tmp, name)], check=True,
                                   capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(name|private_segment_fixed_size|vgpr_count):\s*(\S+)", line)
                if not m:
                    continue
                cur[m.group(1)] = m.group(2)
                if len(cur) == 3:
                    out[cur["name"]] = (int(cur["private_segment_fixed_size"]), int(cur["vgpr_count"]))
                    cur = {}
    return out


@pytest.fixture(scope="module")
def kernels():
    if not (os.path.exists(LIB) and os.path.exists(f"{LLVM}/llvm-readelf")):
        pytest.skip("library or llvm tools not present")
    table = kernel_table()
    assert len(table) > 50, len(table)
    return table


@pytest.mark.parametrize("stem", ["kr_bucket", "kr_part8", "kr_part", "kr_count", "kr_hist", "kr_scan", "kr_finish", "k_pack_updates"])
def test_kernels_that_must_not_spill(kernels, stem):
    found = {k: v for k, v in kernels.items() if re.search(r"\d+" + stem + r"(I|E)", k)}
    assert found, stem
    for name, (private, vgprs) in found.items():
        assert private == 0, (name, private, vgprs)


def test_no_kernel_uses_a_large_private_segment(kernels):
    """Spills of a few registers are tolerated in the fused kernels (they are covered by parity tests at every
    size class; the largest, k_train64 with 32 taps, holds 1.2 KiB per lane); a private segment beyond 2 KiB
    per lane means a local array went to memory."""
    big = {k: v for k, v in kernels.items() if v[0] > 2048}
    assert not big, big

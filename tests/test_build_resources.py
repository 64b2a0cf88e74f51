"""The compiler's resource report of the last build (dsp_amd/csrc/*.res, written by the Makefile): no kernel may use
scratch memory.  Every kernel here is sized against the register file on purpose (launch bounds, occupancy of the
co-running kernels); one that starts spilling has lost a multiple of its speed without failing any numerical test."""
import glob
import os
import re

import pytest

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dsp_amd", "csrc")


def kernel_resources():
    out = {}
    for path in sorted(glob.glob(os.path.join(CSRC, "*.res"))):
        name = None
        for line in open(path, errors="replace"):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                out[name] = {"file": os.path.basename(path)}
                continue
            m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill): (\d+)", line)
            if m and name:
                out[name][m.group(1).split(" [")[0]] = int(m.group(2))
    return out


def test_no_kernel_uses_scratch():
    res = kernel_resources()
    if not res:
        pytest.skip("no resource reports: build with make -C dsp_amd/csrc")
    assert len(res) >= 40, sorted(res)
    # (the one-lane-per-sequence IIR keeps its filter state in accumulation registers by design: "VGPRs Spill" there
    # counts moves to AGPRs, not memory; scratch bytes are what must stay zero)
    spilling = {k: v for k, v in res.items() if v.get("ScratchSize", 0)}
    assert not spilling, spilling
    # the float64 envelope kernel runs four waves per SIMD: it must stay within 128 registers
    env = [v for k, v in res.items() if "corr_env_kernel" in k]
    assert env and env[0]["VGPRs"] <= 128, env
    # so does the fused FIR + correlation + envelope kernel of the full chain (two workgroups of eight waves per unit)
    fce = [v for k, v in res.items() if "fce_kernel" in k]
    assert fce and fce[0]["VGPRs"] <= 128, fce
    # the spectrum + band-signal quality kernel shares a compute unit three workgroups of nine waves at a time
    sh = [v for k, v in res.items() if "spec9_hilbert_kernel" in k]
    assert sh and sh[0]["VGPRs"] <= 72, sh

"""The numerical contract of the HIP library: binary32, one rounding per operation, NO fused
multiply-add contraction.  Checked on the generated gfx950 ISA: every v_fma/v_fmac/v_mad f32 instruction of a kernel
must belong to an IEEE division (5 per v_div_fixup_f32) or square-root (2 per v_sqrt_f32) expansion; the only other
fma users are the float-assisted 64-bit integer divisions (recognisable by their 2^32 constants)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "xna-ray-trace_amd", "csrc")


@pytest.fixture(scope="module")
def isa():
    out = os.path.join(CSRC, "kernels.s")
    subprocess.check_call(["make", "-s", "-C", CSRC, "asm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def kernels(isa_text):
    cur, body = None, {}
    for line in isa_text.split("\n"):
        m = re.match(r"^(_ZN3xrt\w+):", line)
        if m:
            cur = m.group(1)
            body[cur] = []
        elif cur is not None:
            body[cur].append(line)
    return body


def test_no_contracted_fma_in_any_kernel(isa):
    ks = kernels(isa)
    assert any("k_intersect" in k for k in ks) and any("k_shade" in k for k in ks)
    for name, lines in ks.items():
        text = "\n".join(lines)
        assert not re.search(r"\bv_(mad|mac)_f32|\bv_pk_fma_f32|\bv_fma_mix", text), name
        fma = len(re.findall(r"\bv_(?:fma|fmac)_f32", text))
        div = text.count("v_div_fixup_f32")
        sqrt = len(re.findall(r"\bv_sqrt_f32", text))
        # 64-bit integer division helper: v_fmac with 0x4f800000 (2^32) and 0xcf800000 (-2^32)
        idiv = len(re.findall(r"v_fmac_f32_e32 v\d+, 0x4f800000", text)) + len(re.findall(r"v_fmac_f32_e32 v\d+, 0xcf800000", text))
        assert fma == 5 * div + 2 * sqrt + idiv, (name, fma, div, sqrt, idiv)
        fma64 = len(re.findall(r"\bv_(?:fma|fmac)_f64", text))
        div64 = text.count("v_div_fixup_f64")
        sqrt64 = len(re.findall(r"\bv_rsq_f64", text))
        if div64 == 0 and sqrt64 == 0:
            assert fma64 == 0, (name, "double-precision fma outside a division / sqrt expansion", fma64)


def test_hot_kernel_resources(isa):
    """k_intersect must stay within 128 VGPRs without scratch (4 waves per SIMD, MI355X_MICROARCH.md register table).
    Scene mode parks 27 words per lane in LDS, so its deep-stack variants trade a wave for that: (T + 27) KB + 256 B per block."""
    usage = open(os.path.join(CSRC, "kernels.usage.txt")).read()
    blocks = re.findall(r"Function Name: (\S*k_intersectILi(\d+)ELi(\d)E\S*).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)",
                        usage, flags=re.S)
    assert len(blocks) == 15
    for name, cap, mode, vgprs, scratch, occ in blocks:
        want = 4
        if int(mode) == 0 and int(cap) > 12:
            want = 160 * 1024 // ((int(cap) + 27) * 1024 + (512 if int(cap) < 40 else 0))
        assert int(vgprs) <= 128 and int(scratch) == 0 and int(occ) >= want, (name, vgprs, scratch, occ)

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
    subprocess.check_call(["make", "-s", "-C", CSRC, "asm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(os.path.join(CSRC, "kernels.s")).read() + "\n" + open(os.path.join(CSRC, "packet.s")).read()


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


def f64_census(text):
    return {"fma": len(re.findall(r"\bv_(?:fma|fmac)_f64", text)), "fixup": text.count("v_div_fixup_f64"),
            "rsq": len(re.findall(r"\bv_rsq_f64", text)), "rndne": len(re.findall(r"\bv_rndne_f64", text))}


_UNITS = None


def f64_units():
    """f64 fma / marker instructions of ONE double division, square root and remainder() as this compiler expands them
    (a three-kernel probe compiled with the library's flags)."""
    global _UNITS
    if _UNITS is None:
        import tempfile
        flags = re.search(r"^FLAGS\s*:=\s*(.*)$", open(os.path.join(CSRC, "Makefile")).read(), flags=re.M).group(1).split()
        flags = [f for f in flags if f not in ("-fPIC",) and not f.startswith("$(")] + ["--offload-arch=gfx950"]
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "probe.hip")
            open(src, "w").write("#include <hip/hip_runtime.h>\n"
                                 "__global__ void probe_div(double *a) { a[0] = a[1] / a[2]; }\n"
                                 "__global__ void probe_sqrt(double *a) { a[0] = sqrt(a[1]); }\n"
                                 "__global__ void probe_rem(double *a) { a[0] = remainder(a[1], a[2]); }\n")
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["--cuda-device-only", "-S", "-o", os.path.join(d, "probe.s"), src],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            text = open(os.path.join(d, "probe.s")).read()
        parts = {}
        for key in ("div", "sqrt", "rem"):
            m = re.search(r"^_Z\d+probe_%s\w*:(.*?)s_endpgm" % key, text, flags=re.S | re.M)
            parts[key] = f64_census(m.group(1))
        assert parts["div"]["fma"] > 0 and parts["div"]["fixup"] == 1 and parts["sqrt"]["rsq"] == 1 and parts["rem"]["rndne"] > 0, parts
        _UNITS = parts
    return _UNITS


def test_no_contracted_fma_in_any_kernel(isa):
    ks = kernels(isa)
    assert any("k_intersect" in k for k in ks) and any("k_shade" in k for k in ks) and any("k_packet" in k for k in ks)
    for name, lines in ks.items():
        text = "\n".join(lines)
        assert not re.search(r"\bv_(mad|mac)_f32|\bv_pk_fma_f32|\bv_fma_mix", text), name
        fma = len(re.findall(r"\bv_(?:fma|fmac)_f32", text))
        div = text.count("v_div_fixup_f32")
        sqrt = len(re.findall(r"\bv_sqrt_f32", text))
        # 64-bit integer division helper: v_fmac with 0x4f800000 (2^32) and 0xcf800000 (-2^32)
        idiv = len(re.findall(r"v_fmac_f32_e32 v\d+, 0x4f800000", text)) + len(re.findall(r"v_fmac_f32_e32 v\d+, 0xcf800000", text))
        assert fma == 5 * div + 2 * sqrt + idiv, (name, fma, div, sqrt, idiv)
        # double precision (k_shade: SPOT:54 division, RT:676-679 Snell square root, MAT:168-169 Math.IEEERemainder): every
        # f64 fma must belong to one of those library expansions, whose sizes are measured on a probe built with the same flags
        c = f64_census(text)
        u = f64_units()
        n_rem = c["rndne"] // u["rem"]["rndne"]
        n_div = c["fixup"] - n_rem * u["rem"]["fixup"]
        n_sqrt = c["rsq"] - n_rem * u["rem"]["rsq"]
        assert c["rndne"] % u["rem"]["rndne"] == 0 and n_div >= 0 and n_sqrt >= 0, (name, c)
        assert c["fma"] == n_div * u["div"]["fma"] + n_sqrt * u["sqrt"]["fma"] + n_rem * u["rem"]["fma"], (name, "contracted double-precision multiply-add", c, n_div, n_sqrt, n_rem)
        if "k_shade" in name:
            assert (n_div, n_sqrt, n_rem) == (1, 1, 2), (n_div, n_sqrt, n_rem)


def test_hot_kernel_resources(isa):
    """k_intersect's register budget (MI355X_MICROARCH.md register table).  Scene mode: <= 128 VGPRs without scratch, 4 waves per
    SIMD; it parks 27 words per lane in LDS, so its deep-stack variants trade a wave for that: (T + 27) KB + 256 B per block.
    One-body and per-mesh modes: 5 waves per SIMD (96 VGPRs) at the price of at most 12 spilled registers (measured: C5 per-lane
    frames -8 %, the others unchanged); their 40-level variants keep 4 waves (40 KB of LDS stack per block)."""
    usage = open(os.path.join(CSRC, "kernels.usage.txt")).read()
    blocks = re.findall(r"Function Name: (\S*k_intersectILi(\d+)ELi(\d)E\S*).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)",
                        usage, flags=re.S)
    assert len(blocks) == 15
    for name, cap, mode, vgprs, scratch, occ in blocks:
        if int(mode) == 0:
            want, spill = 4, 0
            if int(cap) > 12:
                want = 160 * 1024 // ((int(cap) + 27) * 1024 + (512 if int(cap) < 40 else 0))
        else:
            want, spill = (5, 48) if int(cap) < 40 else (4, 0)
        budget = {5: 96, 4: 128, 3: 168, 2: 256}[want]   # where LDS already limits the waves per SIMD the compiler may use their registers
        assert int(vgprs) <= budget and int(scratch) <= spill and int(occ) >= want, (name, vgprs, scratch, occ)


def test_packet_kernel_argument_offsets():
    """k_packet re-reads the arguments a packet needs once from the kernel-argument segment (packet.hip PkKernarg) at offsets it computes from
    the parameter list: eight pointers, SceneView, PacketArgs.  The code object's own metadata must say the same."""
    asm = open(os.path.join(CSRC, "packet.s")).read()
    src = open(os.path.join(CSRC, "packet.hip")).read()
    assert "PK_KERNARG_SCENE = 8 * 8, PK_KERNARG_ARGS = PK_KERNARG_SCENE + (unsigned)sizeof(SceneView)" in src
    kernels = re.findall(r"\.args:(.*?)\.group_segment_fixed_size:.*?\.name:\s+(\S+)", asm, flags=re.S)
    seen = 0
    for args, name in kernels:
        if "k_packet" not in name:
            continue
        seen += 1
        offs = [(int(o), int(sz), kind) for o, sz, kind in re.findall(r"\.offset:\s+(\d+)\s+\.size:\s+(\d+)\s+\.value_kind:\s+(\w+)", args)]
        ptrs = [x for x in offs if x[2] == "global_buffer"]
        byval = [x for x in offs if x[2] == "by_value"]
        assert [o for o, _, _ in ptrs] == [8 * i for i in range(8)] and len(byval) == 2
        assert byval[0][0] == 64 and byval[1][0] == 64 + byval[0][1]          # SceneView at 64, PacketArgs right behind it (sizeof(SceneView) % 8 == 0)
    assert seen == 5   # the three modes, and the split-walk variants of the two one-body modes


def test_packet_kernel_resources(isa):
    """k_packet: no scratch, and an SGPR allocation within what packet_blocks_per_cu (packet.hip) sizes its grid for -- the
    occupancy API over-reports resident blocks in the 81-112 SGPR range (MI355X_MICROARCH.md, Correctness boundaries)."""
    usage = open(os.path.join(CSRC, "packet.usage.txt")).read()
    src = open(os.path.join(CSRC, "packet.hip")).read()
    waves = int(re.search(r"#define PK_SINGLE_WAVES (\d+)", src).group(1))
    hi, lo = re.search(r"constexpr int PK_SGPRS = PK_SINGLE_WAVES >= 7 \? (\d+) : (\d+);", src).groups()
    budget = int(hi) if waves >= 7 else int(lo)
    blocks = re.findall(r"Function Name: (\S*k_packet\S*).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)", usage, flags=re.S)
    assert len(blocks) == 5   # MODE_SCENE (0), MODE_MESH (1), MODE_SINGLE (2), and the split-walk variants of the one-body modes
    for name, sgprs, vgprs, scratch in blocks:
        mode, sp = (int(x) for x in re.search(r"k_packetILi(\d)ELb(\d)E", name).groups())
        if sp:   # the split-walk variant (off by default) is compiled for six waves per SIMD as well and pays for it with a few dwords of scratch per lane
            assert int(sgprs) <= budget and int(vgprs) <= 80 and int(scratch) <= 64, (name, sgprs, vgprs, scratch)
            continue
        # 112 SGPRs allow six waves per SIMD, and so do up to 80 VGPRs (one-body variants); the two-level variant also carries the
        # scene cursor and the body's transform: four waves per SIMD (128 registers), its scene-level answer parked in LDS
        # (round 3: five waves per SIMD -- 96 registers and a few dwords of scratch per lane that are touched once per packet, measured faster than 4 x 106)
        assert int(sgprs) <= budget and int(vgprs) <= (96 if mode == 0 else 80) and int(scratch) <= (48 if mode == 0 else 0), (name, sgprs, vgprs, scratch)

"""Build-time check of the generated gfx950 code of the production rollout kernels (no GPU needed).

The register-resident weight fragments of `rollout_xdl_kernel` are pinned to AGPRs through inline-asm operand constraints
(cadm_amd/csrc/rollout_xdl.h).  If hipcc ever runs out of registers there it parks a resident fragment in VGPRs / scratch
and copies it back with `v_accvgpr_write` right in front of the asm MFMA that reads it -- an operand hazard the compiler
cannot see through inline asm (stale operands, silently wrong rollouts).  So the shipped library itself is disassembled
and every asm-MFMA geometry must be free of AGPR<->VGPR shuffles and of scratch traffic."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "cadm_amd", "libcadm_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(path):
    """gfx950 ELF images inside the library's .hip_fatbin (clang offload bundles, one per translation unit)."""
    data = open(path, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return out
        n = struct.unpack_from("<Q", data, i + len(MAGIC))[0]
        off = i + len(MAGIC) + 8
        for _ in range(n):
            o, s, ts = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24:off + 24 + ts]
            off += 24 + ts
            if b"gfx950" in triple and s > 0:
                out.append(data[i + o:i + o + s])
        pos = i + len(MAGIC)


def _kernels(image, match="rollout_xdl_kernel"):
    """{demangled-ish symbol: instruction text} of the kernels of one code object whose name contains `match`."""
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(image)
        f.flush()
        txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
    out, name, buf = {}, None, []
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if name is not None:
                out[name] = buf
            name, buf = (m.group(1), []) if match in m.group(1) else (None, [])
        elif name is not None:
            buf.append(line.strip())
    if name is not None:
        out[name] = buf
    return out


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not available")
def test_resident_fragments_never_leave_agprs():
    assert os.path.exists(LIB), "libcadm_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    images = _code_objects(LIB)
    assert len(images) >= 5, "expected the per-env rollout code objects in the library, found %d" % len(images)
    checked = 0
    for img in images:
        for sym, ins in _kernels(img).items():
            # XC<ENV, C, HID>: geometries with HID <= 256 use the asm MFMA path with AGPR-resident fragments
            m = re.search(r"2XCILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi\d+ELi\d+EEE", sym)        # XC<ENV, C, HID, MT, NH, ACT>
            assert m, sym
            hid = int(m.group(3))
            n_mfma = sum(1 for x in ins if x.startswith("v_mfma_f32_16x16x32_f16"))
            assert n_mfma > 30, "%s: only %d f16 MFMAs -- wrong kernel?" % (sym, n_mfma)
            if hid > 256:
                continue
            scratch = [x for x in ins if x.startswith("scratch_")]
            assert not scratch, "%s: %d scratch accesses (e.g. %s)" % (sym, len(scratch), scratch[0])
            # The kernel body exists once per wave variant (tiles per wave); each variant loads ITS resident fragments
            # with a burst of "buffer_load_dwordx4 a[..]" and keeps them to the end.  hipcc may park ordinary VGPR values
            # in OTHER AGPRs (harmless spills); a copy into or out of a resident one is the hazard.
            # variant k = [its prologue][burst of resident loads][tile loop][unconditional branch to the common exit]:
            # a boundary is the first s_branch / s_endpgm behind the last MFMA that precedes the next burst
            # (the asm statement of a resident load is "s_nop 4; buffer_load_dwordx4 a[..]": that pair identifies them --
            #  hipcc may also point ordinary ring loads at spare AGPRs, which it tracks itself)
            loads = [i for i, x in enumerate(ins) if i > 0 and x.startswith("buffer_load_dwordx4 a[") and ins[i - 1].startswith("s_nop 4")]
            assert loads, "%s: no resident-fragment loads" % sym
            mfma_at = [i for i, x in enumerate(ins) if x.startswith("v_mfma_")]
            bursts = []      # a new burst = a resident load with MFMAs between it and the previous burst's loads
            for k, i in enumerate(loads):
                if k == 0 or any(loads[k - 1] < m < i for m in mfma_at):
                    bursts.append(i)
            starts = [0]
            for nb in bursts[1:]:
                last = max(i for i in mfma_at if i < nb)
                end = next(i for i in range(last, nb) if ins[i].startswith(("s_branch", "s_endpgm")))
                starts.append(end + 1)
            for k, b in enumerate(starts):
                e = starts[k + 1] if k + 1 < len(starts) else len(ins)
                resident = set()
                for i in loads:
                    if b <= i < e:
                        m2 = re.match(r"buffer_load_dwordx4 a\[(\d+):(\d+)\]", ins[i])
                        resident.update(range(int(m2.group(1)), int(m2.group(2)) + 1))
                for x in ins[b:e]:
                    m2 = re.match(r"v_accvgpr_write_b32 a(\d+),", x) or re.match(r"v_accvgpr_read_b32 v\d+, a(\d+)", x)
                    assert not (m2 and int(m2.group(1)) in resident), "%s: resident fragment register copied: %s" % (sym, x)
                    m2 = re.match(r"v_mfma_f32_16x16x32_f16 v\[\d+:\d+\], a\[(\d+):(\d+)\], v\[", x)
                    assert not (m2 and not set(range(int(m2.group(1)), int(m2.group(2)) + 1)) <= resident), \
                        "%s: MFMA reads an AGPR operand that is not a resident fragment: %s" % (sym, x)
            # inline-asm loads are invisible to hipcc's hazard recognizer: an SGPR offset written by a VALU instruction
            # (v_readfirstlane / v_readlane) needs 5 wait states before a VMEM instruction reads it
            for i in loads:
                x = ins[i]
                m2 = re.match(r"buffer_load_dwordx4 a\[\d+:\d+\], v\d+, s\[\d+:\d+\], (s\d+) offen", x)
                if not m2:
                    continue
                waits = 0
                for y in reversed(ins[max(0, i - 6):i]):
                    if re.match(r"v_read(first)?lane_b32 %s," % m2.group(1), y):
                        assert waits >= 5, "%s: %s only %d wait states after %s" % (sym, x.split("//")[0], waits, y.split("//")[0])
                        break
                    m3 = re.match(r"s_nop (\d+)", y)
                    waits += int(m3.group(1)) + 1 if m3 else 1
            # every asm MFMA carries its own 2 wait states ("s_nop 1" in the asm statement): whatever VALU instruction hipcc puts
            # in front of it (v_accvgpr_read of a parked operand, a zeroing v_mov), the MFMA never reads a VGPR too early
            for i, x in enumerate(ins):
                if x.startswith("v_mfma_f32_16x16x32_f16"):
                    assert ins[i - 1].startswith("s_nop") and int(ins[i - 1].split()[1]) >= 1, \
                        "%s: MFMA without wait states in front: %s / %s" % (sym, ins[i - 1].split("//")[0], x.split("//")[0])
            res = [x for x in ins if x.startswith("v_mfma_f32_16x16x32_f16") and re.search(r", a\[\d+:\d+\], v\[", x)]
            assert len(res) > 10, "%s: no MFMA reads its A operand from AGPRs -- residency is off" % sym
            checked += 1
    assert checked >= 5 * 2 * 3      # 5 envs x {C = 0, 10} x 3 noise modes at least for HID = 200


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not available")
def test_training_chain_ring_registers_are_left_alone():
    """The training chains' operand ring is a[0:63], named literally in inline asm (cadm_amd/csrc/train.hip): hipcc must
    not touch THOSE registers in chain_kernel once the ring is in use (under VGPR pressure it parks values in AGPRs: the 4-wave flavour,
    cut for three waves per SIMD, does -- above a63, or in the kernel prologue), must not spill to scratch, and no MFMA may directly follow a VALU instruction (the
    asm MFMAs carry no wait states: their operands come from loads and ds_reads)."""
    found = 0
    for img in _code_objects(LIB):
        for sym, ins in _kernels(img, "chain_kernel").items():
            ins = [x.split("//")[0].strip() for x in ins if x.strip()]
            found += 1
            assert not [x for x in ins if x.startswith("scratch_")], "chain_kernel uses scratch"
            mine = ("v_mfma_f32_16x16x4_f32", "global_load_dwordx4 a[")
            # From the first ring load on, a0-a63 hold operand blocks between one asm statement (the load) and another (the MFMAs): a value
            # hipcc parks there would overwrite them.  In front of it -- the kernel prologue: input tiles, stage table, first look-up -- the
            # ring is empty and hipcc may use the registers as it likes (the 4-wave flavour's 84 VGPRs do not hold a whole input tile's
            # loads; the asm statements' clobber lists make it give them up at the first ring load).
            first_ring = min(i for i, x in enumerate(ins) if x.startswith("global_load_dwordx4 a["))
            alien = []
            for x in ins[first_ring:]:
                if x.startswith(mine):
                    continue
                for m in re.finditer(r"\ba\[?(\d+)(?::(\d+))?", x):
                    if int(m.group(1)) < 64:
                        alien.append(x)
            assert not alien, "hipcc touches the ring's AGPRs (a0-a63) behind the first ring load of chain_kernel: %s" % alien[:3]
            mf = [i for i, x in enumerate(ins) if x.startswith("v_mfma_")]
            assert len(mf) >= 64
            for i in mf:
                assert re.match(r"v_mfma_f32_16x16x4_f32 v\[\d+:\d+\], a\d+, v\d+, v\[", ins[i]), ins[i]
                assert ins[i - 1].split()[0].startswith(("s_", "v_mfma")), "VALU instruction in front of an asm MFMA: %s / %s" % (ins[i - 1], ins[i])
            ring = [x for x in ins if x.startswith("global_load_dwordx4 a[")]
            assert ring and all(re.match(r"global_load_dwordx4 a\[\d+:\d+\], v\d+, s\[\d+:\d+\]", x) for x in ring)
    assert found == 2      # chain_kernel<8>, chain_kernel<4>


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not available")
def test_wave_tile_kernel_shape():
    """The wave-tile rollout kernel (cadm_amd/csrc/rollout_wt.h) lives on the edge of the 256 registers two waves per SIMD leave
    it (112-128 of them are the activations it carries from layer to layer).  At the reference's width it must not spill for the
    small and medium observation spaces (halfcheetah, cart-pole, pendulum, ant; slim humanoid's 6 pair slots per lane are known to spill
    a few registers: profiles/r4_wave_tile.md), it must move its weights by LDS-DMA, and a step must hold the 960 MFMAs of 13 + 13 +
    13 + 13 + 3 tiles."""
    seen = 0
    for img in _code_objects(LIB):
        for sym, ins in _kernels(img, "rollout_wt_kernel").items():
            m = re.search(r"2XCILi(\d+)ELi(\d+)ELi(\d+)ELi1ELi(\d+)ELi\d+EEE", sym)
            assert m, sym
            env, ctx, hid = int(m.group(1)), int(m.group(2)), int(m.group(3))
            ins = [x.split("//")[0].strip() for x in ins if x.strip()]
            assert any(x.startswith("buffer_load_dwordx4") and x.endswith(" lds") for x in ins), "%s: no LDS-DMA weight requests" % sym
            n_mfma = sum(1 for x in ins if x.startswith("v_mfma_f32_16x16x32_f16"))
            if hid == 200 and env == 0:
                # layer 0: 13 tiles x (1 or 2) chunks x 3; three hidden layers (the layer loop is unrolled: the activation register
                # sets swap roles): 13 x 7 x 3 each; head: 3 x 7 x 3 -- with two chunks in layer 0 the 960 MFMAs of a rollout step
                nc0 = (18 + 6 + ctx + 31) // 32
                assert n_mfma == 13 * nc0 * 3 + 3 * 13 * 7 * 3 + 3 * 7 * 3, "%s: %d MFMAs" % (sym, n_mfma)
            if hid <= 200 and env != 2:
                scratch = [x for x in ins if x.startswith("scratch_")]
                assert not scratch, "%s: %d scratch accesses (e.g. %s)" % (sym, len(scratch), scratch[0])
            # the weight ring is only correct if every wave's LDS-DMA pieces (tracked by vmcnt) have landed BEFORE the block
            # barrier -- walking back from every s_barrier, a `s_waitcnt vmcnt(0)` must come before any LDS-DMA request
            for i, x in enumerate(ins):
                if not x.startswith("s_barrier"):
                    continue
                j = i - 1
                while j >= 0 and not (ins[j].startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", ins[j])):
                    assert not (ins[j].startswith("buffer_load_dwordx4") and ins[j].endswith(" lds")), \
                        "%s: an LDS-DMA request at instruction %d reaches the barrier at %d without a vmcnt(0) wait" % (sym, j, i)
                    assert not ins[j].startswith(("s_barrier", "s_endpgm")), "%s: barrier at %d is not preceded by s_waitcnt vmcnt(0)" % (sym, i)
                    j -= 1
                assert j >= 0, "%s: barrier at %d is not preceded by s_waitcnt vmcnt(0)" % (sym, i)
            seen += 1
    assert seen >= 5 * 2 * 3


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not available")
def test_coissue_bench_streams_are_what_they_claim(tmp_path):
    """VERDICT r4 #1a: the MFMA / VALU co-issue micro-benchmark of rounds 3-4 was SLP-packed (542 v_pk_*_f32) and clustered by hipcc, so it
    measured packed fp32 next to MFMAs -- which does not co-execute -- and the conclusion drawn from it ("VALU and MFMA do not overlap on
    gfx950") was wrong for plain VALU.  tools/micro/gen_coissue_bench.py writes every stream as ONE asm block; this test compiles a sample of
    its variants and checks the disassembly: exactly F fillers behind every MFMA, evenly spaced, no packed fp32 in the scalar variants and
    nothing but packed fp32 in the packed ones."""
    import importlib.util
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    spec = importlib.util.spec_from_file_location("gen_coissue_bench", os.path.join(ROOT, "tools", "micro", "gen_coissue_bench.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    sample = [v for v in gen.variants() if v[0] in ("solo_m16_n4_fma3", "both_m16_n8_fma2", "solo_m32_n2_fma5", "solo_m16_n4_pkfma2", "both_m16_n4_epi3",
                                                     "both_m16_n4_epipk3", "pair_m16_n4_fma")]
    assert len(sample) == 7
    src = ["#include <hip/hip_runtime.h>", "typedef float floatx2 __attribute__((ext_vector_type(2)));", "typedef float floatx4 __attribute__((ext_vector_type(4)));",
           "typedef float floatx16 __attribute__((ext_vector_type(16)));", "typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));"]
    src += [gen.variant(*v) for v in sample]
    hip = tmp_path / "coissue_sample.hip"
    hip.write_text("\n".join(src))
    obj = tmp_path / "coissue_sample.o"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-c", str(hip), "-o", str(obj)], check=True, capture_output=True)
    kernels = {}
    for img in _code_objects(str(obj)):
        for sym, ins in _kernels(img, "k_").items():
            kernels[sym] = [x.split("//")[0].strip() for x in ins if x.strip()]
    for (vname, mf, nm, kind, F, threads, pair) in sample:
        ins = next(v for k, v in kernels.items() if ("k_" + vname) in k)
        mf_at = [i for i, x in enumerate(ins) if x.startswith(gen.MFMA[mf][0])]
        assert len(mf_at) == nm, "%s: %d MFMAs in the loop body, expected %d" % (vname, len(mf_at), nm)
        packed = [x for x in ins[mf_at[0]:mf_at[-1] + F + 1] if x.startswith("v_pk_")]
        if pair:      # the MFMA stream is bare; the filler stream is a second loop
            assert all(b - a == 1 for a, b in zip(mf_at, mf_at[1:])), vname
            continue
        # evenly spaced: exactly F fillers between consecutive MFMAs, and behind the last one
        assert all(b - a == F + 1 for a, b in zip(mf_at, mf_at[1:])), "%s: fillers are not evenly spaced" % vname
        body = ins[mf_at[0]:mf_at[-1] + F + 1]
        fillers = [x for i, x in enumerate(body) if (i % (F + 1)) != 0]
        assert len(fillers) == nm * F
        if kind in ("fma", "epi"):
            assert not packed, "%s: packed fp32 in a scalar stream: %s" % (vname, packed[:2])
            if kind == "fma":
                assert all(x.startswith("v_fma_f32") for x in fillers), vname
        if kind == "pkfma":
            assert all(x.startswith("v_pk_fma_f32") for x in fillers), vname
        if kind == "epipk":
            assert packed, vname

"""Build-time check of the generated gfx950 code of the production kernels (no GPU needed).

The register-resident weight fragments of `rollout_xdl_kernel` are pinned to AGPRs through inline-asm operand constraints
(cadm_amd/csrc/rollout_xdl.h).  If hipcc ever runs out of registers there it parks a resident fragment in VGPRs / scratch
and copies it back with `v_accvgpr_write` right in front of the asm MFMA that reads it -- an operand hazard the compiler
cannot see through inline asm (stale operands, silently wrong rollouts).  So the shipped library itself is disassembled
and every asm-MFMA geometry must be free of AGPR<->VGPR shuffles and of scratch traffic.

The rules themselves live in the product (cadm_amd/isa_check.py): `cadm_amd.jit.build` applies them to every module it compiles on the
user's machine, with that machine's hipcc.  This file applies them to the shipped library and adds what is specific to
the compiled-in geometries (MFMA counts, no scratch at the reference's widths, number of kernels)."""
import os
import re
import subprocess

import pytest

from cadm_amd import isa_check

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "cadm_amd", "libcadm_hip.so")
OBJDUMP = isa_check.find_objdump("/opt/rocm/bin/hipcc") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
_code_objects = isa_check.code_objects


def _kernels(image, match="rollout_xdl_kernel"):
    return isa_check.kernels(image, match, OBJDUMP)


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
            problems, info = isa_check.check_rollout_xdl(sym, ins)      # R1 resident AGPRs, R2 readfirstlane wait states, R3 s_nop in front of asm MFMAs
            assert info["mfma"] > 30, "%s: only %d f16 MFMAs -- wrong kernel?" % (sym, info["mfma"])
            assert not problems, problems[:3]
            if hid > 256:
                continue
            assert info["scratch"] == 0, "%s: %d scratch accesses" % (sym, info["scratch"])
            assert info["resident_loads"] > 0, "%s: no resident-fragment loads" % sym
            assert info["resident_mfma"] > 10, "%s: no MFMA reads its A operand from AGPRs -- residency is off" % sym
            checked += 1
    assert checked >= 5 * 2 * 3      # 5 envs x {C = 0, 10} x 3 noise modes at least for HID = 200


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not available")
def test_training_chain_ring_registers_are_left_alone():
    """The training chains' operand ring is a[0:63], named literally in inline asm (cadm_amd/csrc/train.hip): hipcc must
    not touch THOSE registers in chain_kernel once the ring is in use (under VGPR pressure it parks values in AGPRs: the 4-wave flavour,
    cut for three waves per SIMD, does -- above a63, or in the kernel prologue), must not spill to scratch, and no MFMA may directly follow a VALU instruction (the
    asm MFMAs carry no wait states: their operands come from loads and ds_reads).  "Once the ring is in use" follows the kernel's control
    flow from the first ring load (isa_check.reachable_from), not the linear order of the image."""
    found = 0
    for img in _code_objects(LIB):
        for sym, ins in _kernels(img, "chain_kernel").items():
            found += 1
            problems, info = isa_check.check_chain(sym, ins)
            assert not problems, problems[:3]
            assert info["scratch"] == 0, "chain_kernel uses scratch"
            assert info["mfma"] >= 64
            # the reachability is a real restriction (the prologue is excluded) and a real graph (the loops are included)
            assert info["reachable"] > 1000 and info["before_ring_only"] > 0, info
    assert found == 2      # chain_kernel<8>, chain_kernel<4>


def test_reachability_follows_branches_not_layout():
    """isa_check.reachable_from on a hand-made listing: a block placed BEFORE the start instruction in the image but branched to from
    behind it is reachable; the straight-line prologue is not."""
    def line(addr, text, tgt=None):
        return "%-40s // %012X: 00000000%s" % (text, addr, " <k+0x%x>" % (tgt - 0x1000) if tgt is not None else "")
    ins = [line(0x1000, "s_mov_b32 s0, 0"),                       # 0 prologue
           line(0x1004, "s_branch 2", 0x1010),                    # 1 -> 4
           line(0x1008, "v_accvgpr_write_b32 a3, v1"),            # 2 loop tail, placed early in the image
           line(0x100c, "s_branch 1", 0x1014),                    # 3 -> 5
           line(0x1010, "global_load_dwordx4 a[0:3], v2, s[0:1]"),    # 4 START
           line(0x1014, "v_mfma_f32_16x16x4_f32 v[0:3], a0, v4, v[0:3]"),      # 5
           line(0x1018, "s_cbranch_scc1 65531", 0x1008),          # 6 -> 2
           line(0x101c, "s_endpgm")]                              # 7
    assert isa_check.reachable_from(ins, 4) == {2, 3, 4, 5, 6, 7}
    problems, _ = isa_check.check_chain("k", ins)
    assert any("T1" in p and "a3" in p for p in problems), problems


def test_dominance_separates_the_wave_variants():
    """isa_check.dominated_by: hipcc merges the tails of the rollout kernel's wave variants into a shared block guarded by a flag, so variant B
    is REACHABLE from variant A's resident loads although no wave ever goes that way; it is not DOMINATED by them.  (This is what kept the
    scan from flagging the legitimate AGPR spills of a variant without resident fragments -- one hidden layer, built by the GPU box's hipcc.)"""
    def line(addr, text, tgt=None):
        return "%-40s // %012X: 00000000%s" % (text, addr, " <k+0x%x>" % (tgt - 0x2000) if tgt is not None else "")
    ins = [line(0x2000, "s_cmp_lt_u32 s3, 5"),                                  # 0 entry
           line(0x2004, "s_cbranch_scc0 7", 0x2024),                            # 1 -> 9: variant B's entry
           line(0x2008, "s_nop 4"),                                             # 2 variant A
           line(0x200c, "buffer_load_dwordx4 a[4:7], v1, s[8:11], s2 offen"),   # 3 A's resident load
           line(0x2010, "s_nop 1"),                                             # 4
           line(0x2014, "v_mfma_f32_16x16x32_f16 v[0:3], a[4:7], v[8:11], v[0:3]"),      # 5
           line(0x2018, "s_cbranch_scc1 65531", 0x2008),                        # 6 A's tile loop
           line(0x201c, "s_mov_b64 s[4:5], 0"),                                 # 7 A's tail: flag = 0, falls into the shared block
           line(0x2020, "s_branch 1", 0x2028),                                  # 8 -> 10
           line(0x2024, "s_mov_b64 s[4:5], -1"),                                # 9 B's entry: flag = 1
           line(0x2028, "s_cbranch_vccz 2", 0x2034),                            # 10 shared: flag ? variant B : exit
           line(0x202c, "v_accvgpr_write_b32 a5, v9"),                          # 11 variant B parks a value in a5: legitimate
           line(0x2030, "v_accvgpr_read_b32 v9, a5"),                           # 12
           line(0x2034, "s_endpgm")]                                            # 13
    assert {11, 12} <= isa_check.reachable_from(ins, 3)
    assert isa_check.dominated_by(ins, 3) == {3, 4, 5, 6, 7, 8}
    problems, info = isa_check.check_rollout_xdl("k", ins)
    assert not [p for p in problems if "copied" in p], problems
    bad = list(ins)
    bad[4] = line(0x2010, "v_accvgpr_read_b32 v9, a5")                          # the same copy INSIDE variant A is the hazard
    problems, _ = isa_check.check_rollout_xdl("k", bad)
    assert any("R1 resident fragment register copied" in p for p in problems), problems


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
            problems, info = isa_check.check_rollout_wt(sym, ins)      # W1 LDS-DMA, W2 vmcnt(0) in front of every block barrier
            assert not problems, problems[:3]
            if hid == 200 and env == 0:
                # layer 0: 13 tiles x (1 or 2) chunks x 3; three hidden layers (the layer loop is unrolled: the activation register
                # sets swap roles): 13 x 7 x 3 each; head: 3 x 7 x 3 -- with two chunks in layer 0 the 960 MFMAs of a rollout step
                nc0 = (18 + 6 + ctx + 31) // 32
                assert info["mfma"] == 13 * nc0 * 3 + 3 * 13 * 7 * 3 + 3 * 7 * 3, "%s: %d MFMAs" % (sym, info["mfma"])
            if hid <= 200:
                assert info["scratch"] == 0, "%s: %d scratch accesses" % (sym, info["scratch"])
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


# ------------------------------------------------------------------------------------------------------------------
# the same rules on what cadm_amd.jit builds
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not available")
def test_jit_build_scans_the_module_it_compiled(tmp_path, monkeypatch):
    """jit.build cross-compiles a geometry the library does not carry (hidden 144 x 2, context 7, relu), scans the fresh .so with the
    compiler's own llvm-objdump, writes the report next to the module -- and REFUSES a module that breaks a rule (nothing is left in
    the cache for a later call to load)."""
    import json
    from cadm_amd import _lib, jit
    if not jit.hipcc():
        pytest.skip("hipcc not available")
    monkeypatch.setenv("CADM_JIT_CACHE", str(tmp_path))
    monkeypatch.setattr(jit, "_memo", {})
    key = (0, 7, 144, 2, _lib.ACT_KINDS["relu"], 0)
    path = jit.build(*key)
    assert os.path.exists(path)
    rep = json.load(open(path + ".isa.json"))
    assert rep["checked"] and rep["problems"] == 0 and rep["kernels"] >= 3, rep      # cooperative one / two row tiles + wave-tile
    assert os.path.basename(rep["objdump"]) == "llvm-objdump"
    # an independent scan of the cached module agrees
    assert isa_check.scan(path, which=("rollout_xdl_kernel", "rollout_wt_kernel"))["problems"] == []
    # a module that breaks a rule is not registered: simulate a compiler that copied a resident fragment
    real = isa_check.check_rollout_xdl

    def broken(sym, raw):
        problems, info = real(sym, raw)
        return problems + ["%s: R1 resident fragment register copied: v_accvgpr_read_b32 v9, a17 (simulated)" % sym], info
    monkeypatch.setattr(isa_check, "CHECKS", (("rollout_xdl_kernel", broken),) + isa_check.CHECKS[1:])
    key2 = key[:5] + (2,)
    with pytest.raises(_lib.CadmError, match="hand-scheduling rules"):
        jit.build(*key2)
    assert not os.path.exists(jit.module_path(*key2))
    assert not [f for f in os.listdir(tmp_path) if ".tmp" in f]


@pytest.mark.gpu
def test_modules_built_on_this_box_were_scanned_and_the_loaded_library_is_clean(gpu):
    """On the GPU box the JIT runs the BOX's hipcc (a different ROCm point release than the one that built libcadm_hip.so): every module
    it registered must carry a clean scan report made with that toolchain's llvm-objdump, and the library the process has actually
    mapped passes the rules too."""
    import json
    import numpy as np
    from cadm_amd import _lib, jit, synth
    from helpers import make_engine
    prob = synth.make_problem(env="halfcheetah", context=True, E=5, m=1, H=2, hidden_sizes=(144,) * 2, C=7, seed=3)
    eng = make_engine(prob, p=5, H=2, hidden_nonlinearity="relu")
    assert not eng.lib.cadm_rollout_builtin(eng._ctx)
    plan = eng.cem_plan(prob["obs"], prob["cp_obs"], prob["cp_act"], np.zeros((1, 2, 6)), np.full((1, 2, 6), 0.25), 64, seed=1, call=1)
    assert np.isfinite(plan.cpu().numpy()).all()
    mods = [p for p in jit._loaded if "_c7_h144_n2_" in p]
    assert mods
    objdump = isa_check.find_objdump(jit.hipcc())
    assert objdump, "no llvm-objdump next to %s: the JIT could not scan what it built" % jit.hipcc()
    for p in mods:
        rep = json.load(open(p + ".isa.json"))
        assert rep["checked"] and rep["problems"] == 0 and rep["kernels"] >= 2, (p, rep)
    eng.close()
    lib_path = next(l.split()[-1] for l in open("/proc/self/maps") if l.rstrip().endswith("libcadm_hip.so"))
    rep = isa_check.scan(lib_path, objdump)
    assert rep["kernels"] >= 100 and rep["problems"] == [], rep["problems"][:3]

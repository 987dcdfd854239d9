#!/usr/bin/env python3
"""Instruction ledger of the wave-tile rollout kernel (rollout_wt_kernel, halfcheetah / context 10 / hidden 200 / device noise) read off the
shipped code object: what ONE wave issues per rollout STEP, split by phase.   python tools/valu_ledger.py [lib.so ...]
(static counts inside the step loop; the loop has no inner loops, exec-masked regions are counted as issued)"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cadm_amd import isa_check as ic

KERNEL = "rollout_wt_kernel"
GEO = "XCILi0ELi10ELi200ELi1ELi4ELi0EEELi0E"      # XC<halfcheetah, C = 10, HID = 200, MT = 1, NH = 4, swish>, NOISE = Philox


def cls(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_sin", "v_cos", "v_rsq")):
        return "valu_trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("buffer_", "global_")):
        return "vmem"
    return "other"


def ledger(path):
    od = ic.find_objdump()
    for img in ic.code_objects(path):
        for sym, raw in ic.kernels(img, KERNEL, od).items():
            if GEO not in sym:
                continue
            ins = ic.strip(raw)
            addr = [ic._addr(x) for x in raw]
            at = {a: i for i, a in enumerate(addr)}
            base = addr[0]
            mf = [i for i, x in enumerate(ins) if x.startswith("v_mfma")]
            # the step loop: the far backward branch behind the last MFMA, and its target
            end = start = None
            for i in range(mf[-1], len(ins)):
                if ins[i].startswith("s_branch") or ins[i].startswith("s_cbranch"):
                    m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", raw[i])
                    t = at.get(base + int(m.group(1), 16)) if m else None
                    if t is not None and t < mf[0]:
                        end, start = i, t
                        break
            regions = collections.OrderedDict([("state phase (head, noise, state update, reward, input assembly)", (start, mf[0] - 1)),
                                               ("dense layers (MFMAs, fragment reads, block boundaries, tile epilogues)", (mf[0] - 1, end + 1))])
            out = collections.OrderedDict()
            for name, (a, b) in regions.items():
                out[name] = collections.Counter(cls(x.split()[0]) for x in ins[a:b] if x)
            ops = collections.Counter(x.split()[0] for x in ins[start:mf[0] - 1] if x)
            epi = collections.Counter(x.split()[0] for x in ins[mf[0] - 1:end + 1] if x.startswith("v_") and not x.startswith("v_mfma"))
            return dict(sym=sym, total=len(ins), loop=(start, end), regions=out, state_ops=ops, layer_valu_ops=epi, n_mfma=len(mf))
    return None


if __name__ == "__main__":
    libs = sys.argv[1:] or [os.path.join(ROOT, "cadm_amd", "libcadm_hip.so")]
    for lib in libs:
        L = ledger(lib)
        print("## %s" % os.path.relpath(lib, ROOT))
        print("kernel instructions %d, step loop = instructions %d..%d, %d MFMAs per step\n" % (L["total"], L["loop"][0], L["loop"][1], L["n_mfma"]))
        keys = ["mfma", "valu", "valu_trans", "lds", "vmem", "salu", "waitcnt", "branch"]
        print("| phase | " + " | ".join(keys) + " | VALU (incl. transcendental) per MFMA |")
        print("|---|" + "---|" * (len(keys) + 1))
        tot = collections.Counter()
        for name, c in L["regions"].items():
            tot.update(c)
            print("| %s | " % name + " | ".join(str(c[k]) for k in keys) + " | |")
        print("| **step** | " + " | ".join(str(tot[k]) for k in keys) + " | **%.2f** |" % ((tot["valu"] + tot["valu_trans"]) / tot["mfma"]))
        v = sum(L["layer_valu_ops"].values())
        print("\ntile epilogues: %d VALU instructions in the dense layers = %.1f per hidden / head tile (52 + 3 tiles per step): %s"
              % (v, v / 55.0, ", ".join("%s %d" % kv for kv in L["layer_valu_ops"].most_common(10))))
        print("state phase, most frequent: %s\n" % ", ".join("%s %d" % kv for kv in L["state_ops"].most_common(14)))

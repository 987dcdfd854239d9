#!/usr/bin/env python3
"""Developer report on the generated code of the xdl rollout kernels in an object file / library:
per kernel instruction mix and the AGPR<->VGPR / scratch / lane-spill traffic that should not be there.
usage: tools/isa_report.py [file.o | libcadm_hip.so]"""
import collections
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_isa_hygiene as t  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else t.LIB
for img in t._code_objects(path):
    for sym, ins in t._kernels(img).items():
        ops = collections.Counter(x.split()[0] for x in ins if x and not x.startswith(("//", ";")))
        m = re.search(r"2XCILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi\d+ELi\d+EEELi(\d+)", sym)
        key = ("env%s C%s HID%s MT%s noise%s" % m.groups()) if m else sym
        pick = {k: ops[k] for k in ("v_mfma_f32_16x16x32_f16", "v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_writelane_b32",
                                    "v_readlane_b32", "scratch_load_dwordx4", "scratch_store_dwordx4", "buffer_load_dwordx4",
                                    "ds_read_b128", "s_waitcnt", "s_barrier", "s_nop")}
        print("%-28s n=%6d  %s" % (key, len(ins), "  ".join("%s=%d" % (k.replace("v_mfma_f32_16x16x32_f16", "mfma").replace("_b32", "").replace("_dwordx4", ""), v) for k, v in pick.items() if v)))

if len(sys.argv) > 2:      # positions of selected ops of one kernel: tools/isa_report.py file noise2 v_readlane
    for img in t._code_objects(path):
        for sym, ins in t._kernels(img).items():
            if ("EEELi%sE" % sys.argv[2][-1]) not in sym:
                continue
            marks = []
            for i, x in enumerate(ins):
                op = x.split()[0] if x else ""
                if op in ("s_barrier",) or any(op.startswith(q) for q in sys.argv[3:]) or "s_cbranch" in op:
                    marks.append("%d:%s" % (i, op.replace("s_cbranch_", "br_")))
            print(sym[-40:], " ".join(marks))

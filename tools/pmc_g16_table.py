#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per generator kernel template over the LAST bench step of each pass and print
derived ratios (matrix-pipe busy, LDS conflicts, L2 hit rate, L1->L2 request bytes).
usage: pmc_g16_table.py <counter_collection.csv> [more.csv ...]"""
import collections
import csv
import re
import sys

FAM = re.compile(r"(g16_convp|g16_conv|g16_pair|g16_rw|g16_rc|g16_pp|g16_chain|g16_ups|conv1d_f32_mfma|conv_frame_f16s|conv_frame_splitk|attn_relpos_f16s|attn_pack_f16s)<([^>]*)>")
MANGLED = re.compile(r"_ZN3vsp\d+(attn_relpos_f16s|attn_pack_f16s)I((?:Li[0-9]+E)+)E")
tot = collections.OrderedDict()   # family -> counter -> sum
cnt = collections.Counter()
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    # the step boundary: dispatch ids of the first generator family repeat once per step (warm-up + timed); keep the last half
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    half = ids[len(ids) // 2] if ids else 0
    seen = set()
    for r in rows:
        name = r["Kernel_Name"]
        ma = MANGLED.search(name)          # (rocprofv3 leaves some template instantiations mangled)
        if ma:
            name = f"{ma.group(1)}<{', '.join(x[2:-1] for x in re.findall(r'Li[0-9]+E', ma.group(2)))}>"
        m = FAM.search(name)
        if not m or int(r["Dispatch_Id"]) < half:
            continue
        fam = f"{m.group(1)}<{m.group(2).replace(' ', '')}>"
        d = tot.setdefault(fam, collections.OrderedDict())
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if (path, r["Dispatch_Id"]) not in seen and path == sys.argv[1]:
            seen.add((path, r["Dispatch_Id"]))
            cnt[fam] += 1

def g(d, k):
    return d.get(k, float("nan"))

for fam, d in tot.items():
    print(f"== {fam}   ({cnt[fam]} launches in the step)")
    for k, v in d.items():
        print(f"   {k:34s} {v:14.4g}")
    wc = g(d, "SQ_WAVE_CYCLES"); busy = g(d, "SQ_BUSY_CYCLES"); gui = g(d, "GRBM_GUI_ACTIVE")
    print("   -- derived")
    print(f"   wave cycles: waiting (s_waitcnt / barrier) {g(d,'SQ_WAIT_ANY')/wc:.1%}, issue-stalled {g(d,'SQ_WAIT_INST_ANY')/wc:.1%}, issuing {g(d,'SQ_ACTIVE_INST_ANY')/wc:.1%}")
    print(f"   MFMA busy cycles / SQ busy cycles (per SE aggregate): {g(d,'SQ_VALU_MFMA_BUSY_CYCLES')/busy:.3f}")
    print(f"   MFMA f16 MOPS (x512 FLOP): {g(d,'SQ_INSTS_VALU_MFMA_MOPS_F16'):.4g} -> {g(d,'SQ_INSTS_VALU_MFMA_MOPS_F16')*512/1e12:.3f} TFLOP issued")
    print(f"   LDS: instructions {g(d,'SQ_INSTS_LDS'):.4g}, array-active cycles {g(d,'SQ_LDS_IDX_ACTIVE'):.4g}, bank-conflict cycles {g(d,'SQ_LDS_BANK_CONFLICT'):.4g} ({g(d,'SQ_LDS_BANK_CONFLICT')/max(g(d,'SQ_LDS_IDX_ACTIVE'),1):.1%} of active)")
    hit, miss = g(d, "TCC_HIT_sum"), g(d, "TCC_MISS_sum")
    print(f"   L2: requests {g(d,'TCC_REQ_sum'):.4g} (reads {g(d,'TCC_READ_sum'):.4g}), hit rate {hit/(hit+miss):.3f}")
    print(f"   L1->L2 read requests {g(d,'TCP_TCC_READ_REQ_sum'):.4g} (x64 B = {g(d,'TCP_TCC_READ_REQ_sum')*64/1e9:.2f} GB; x128 B = {g(d,'TCP_TCC_READ_REQ_sum')*128/1e9:.2f} GB), write requests {g(d,'TCP_TCC_WRITE_REQ_sum'):.4g}")

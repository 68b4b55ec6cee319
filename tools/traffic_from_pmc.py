#!/usr/bin/env python3
"""HBM bytes per launch of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE collected separately, as MI355X_MICROARCH.md prescribes: FETCH_SIZE takes 3 of the 4 TCC
slots, WRITE_SIZE 2).  Units are KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a
wide (16 B/lane) coalesced streaming read, so the read side is doubled; WRITE_SIZE is taken as is.
usage: traffic_from_pmc.py <kernel substr[|substr...]> <fetch counter_collection.csv> <write counter_collection.csv> <generator> <utterances> [padded_frames] [kernels tag] [steps in the pmc run]"""
import csv, hashlib, json, os, sys
sub, fcsv, wcsv, gen, utt = sys.argv[1:6]
LIB = os.environ.get("VSP_LIB_PATH") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vispeech_amd", "lib", "libvispeech_hip.so")
lib_sha = hashlib.sha256(open(LIB, "rb").read()).hexdigest()   # bench.py reports these bytes only for THIS library build
frames = int(sys.argv[6]) if len(sys.argv) > 6 else 489
steps = int(sys.argv[8]) if len(sys.argv) > 8 else 2   # bench.py --steps 1 --warmup 1 --profile-steps 0
ktag = sys.argv[7] if len(sys.argv) > 7 else 'g16c1'   # g16 kernels, ResBlock chains per VSP_CHAIN=1 (the default)
def total(path, counter):
    s = 0.0; n = 0
    for r in csv.DictReader(open(path)):
        if any(x in r['Kernel_Name'] for x in sub.split('|')) and r['Counter_Name'] == counter:
            s += float(r['Counter_Value']); n += 1
    return s, n
f, nf = total(fcsv, 'FETCH_SIZE')
w, nw = total(wcsv, 'WRITE_SIZE')
assert nf == nw and nf > 0, (nf, nw)
fetch_b, write_b = 2.0 * f * 1024.0, w * 1024.0
out = {"kernel": sub, "kernels": ktag, "lib_sha256": lib_sha, "generator": gen, "utterances": int(utt), "padded_frames": frames, "launches": nf, "launches_per_step": nf / steps,
       "fetch_bytes_per_launch": fetch_b / nf, "write_bytes_per_launch": write_b / nf,
       "hbm_bytes_per_launch": (fetch_b + write_b) / nf,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), KiB units, FETCH_SIZE x2 (gfx950 "
                 "wide-read correction, MI355X_MICROARCH.md section HBM)"}
# per kernel FAMILY (bench.py's dominant_kernel reads the entry of its family): the fused-pair / chain kernels have names of
# their own; the per-convolution kernel serves several families and stays one entry
fams = {}
for key, pat in (("pair64", "g16_pair<"), ("pair32", "g16_rw<"), ("chain32", "g16_rc<"), ("pair128", "g16_pp<"),
                 ("conv_all_widths", "g16_conv"), ("ups_k4", "g16_ups<")):
    def tot(path, counter):
        s2 = 0.0; n2 = 0
        for r in csv.DictReader(open(path)):
            if pat in r['Kernel_Name'] and r['Counter_Name'] == counter:
                s2 += float(r['Counter_Value']); n2 += 1
        return s2, n2
    ff, n1 = tot(fcsv, 'FETCH_SIZE')
    ww, n2 = tot(wcsv, 'WRITE_SIZE')
    if n1 and n1 == n2:
        fams[key] = {"kernel": pat, "launches_per_step": n1 / steps, "fetch_bytes_per_launch": 2.0 * ff * 1024.0 / n1,
                     "write_bytes_per_launch": ww * 1024.0 / n1, "hbm_bytes_per_launch": (2.0 * ff + ww) * 1024.0 / n1}
out["families"] = fams
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""ISA census of the middle scan's hot loop: instructions of one trip of k_mid_flat<2, Hot>'s inner loop
(4 chunks = 64 columns, two adapters) by class, per 16-column chunk and per column pair, from the compiler's assembly
(make -C tgsfilter_amd/csrc asm -> /tmp/tgsf_lib.s), next to the PMC count of a launch (profiles/traffic.json) when given.
  python3 tools/scan_census.py [/tmp/tgsf_lib.s] [profiles/traffic.json] > profiles/r04_scan_isa_census.txt"""
import collections, json, re, sys
asm = sys.argv[1] if len(sys.argv) > 1 else "/tmp/tgsf_lib.s"
traffic = sys.argv[2] if len(sys.argv) > 2 else None
KERNEL = "_ZN4tgsf10k_mid_flatILi2ENS_3HotELi0EEEvNS_9DevParamsENS_8DevBatchEii"
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
# loops: a label followed by an 'Inner Loop Header' comment; its last back edge closes it
best = None
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if not m:
        continue
    hdr = "\n".join(body[i:i + 4])
    if "Inner Loop Header" not in hdr:
        continue
    lab = m.group(1)
    back = [j for j in range(i + 1, len(body)) if re.search(r"\bs_c?branch\w*\s+" + re.escape(lab) + r"$", body[j].strip())]
    if not back:
        continue
    region = body[i:back[-1] + 1]
    n = sum(1 for x in region if "v_lshl_add_u64" in x)
    if best is None or n > best[0]:
        best = (n, lab, region)
n_add, lab, region = best
ins = [x.strip().split()[0] for x in region if x.startswith("\t") and not x.strip().startswith((";", "."))]
chunks = sum(1 for o in ins if o == "v_bitop3_b32") / 64.0  # two v_bitop3 per adapter and column: 64 per 16-column chunk
CLASSES = [
    ("column: v_and_b32 (Eq & Pv)", lambda o: o == "v_and_b32_e32"),
    ("column: v_lshl_add_u64 (sum; + one per fetched chunk: its address)", lambda o: o == "v_lshl_add_u64"),
    ("column: v_or3_b32", lambda o: o == "v_or3_b32"),
    ("column: v_bfi_b32", lambda o: o == "v_bfi_b32"),
    ("column: v_lshlrev_b64 (Ph, Mh << 1)", lambda o: o == "v_lshlrev_b64"),
    ("column: v_bitop3_b32 (Mv')", lambda o: o == "v_bitop3_b32"),
    ("Eq address: v_lshlrev_b32_sdwa (byte of the text dword -> LDS offset)", lambda o: o.startswith("v_lshlrev_b32")),
    ("4th-column test: v_bcnt_u32_b32", lambda o: o == "v_bcnt_u32_b32"),
    ("4th-column test + loop / event tests: v_cmp_*", lambda o: o.startswith("v_cmp")),
    ("register copies: v_mov_b32 / v_mov_b64", lambda o: o.startswith("v_mov")),
    ("other VALU (column counter, fetch offsets, stretch bookkeeping)", lambda o: o.startswith("v_")),
    ("LDS: ds_read_b64 (Eq rows)", lambda o: o.startswith("ds_")),
    ("VMEM: global_load_dwordx4 (text, 16 bytes)", lambda o: o.startswith("global_") or o.startswith("buffer_") or o.startswith("flat_")),
    ("SALU (exec masks of the rare paths, branches, s_waitcnt, s_nop)", lambda o: o.startswith("s_")),
]
cnt = collections.OrderedDict((name, 0) for name, _ in CLASSES)
detail = collections.defaultdict(collections.Counter)
for o in ins:
    for name, f in CLASSES:
        if f(o):
            cnt[name] += 1
            detail[name][o] += 1
            break
valu = sum(v for k, v in cnt.items() if not k.startswith(("LDS", "VMEM", "SALU")))
core = sum(v for k, v in cnt.items() if k.startswith("column"))
print("ISA census of the middle scan's hot loop: k_mid_flat<2, tgsf::Hot> (two adapters of 33..64 bp per pass), gfx950")
print("source: hipcc -O3 --offload-arch=gfx950 -S (make -C tgsfilter_amd/csrc asm); inner loop %s, %d instructions a trip = %.0f chunks of 16 columns" % (lab, len(ins), chunks))
print()
print("%-86s %9s %9s %9s" % ("class", "per trip", "per chunk", "per col."))
for k, v in cnt.items():
    print("%-86s %9d %9.2f %9.3f" % (k, v, v / chunks, v / chunks / 16))
    if k.startswith(("other", "SALU", "4th-column test +")):
        print("    " + ", ".join("%s x%d" % kv for kv in detail[k].most_common(12)))
print()
print("VALU instructions per column pair (both adapters): %.2f   of which the Myers column itself: %.2f (17 per adapter)" % (valu / chunks / 16, core / chunks / 16))
print("  Eq address %.2f, 4th-column test %.2f (v_bcnt) + compares, copies %.2f, rest %.2f" % (
    cnt["Eq address: v_lshlrev_b32_sdwa (byte of the text dword -> LDS offset)"] / chunks / 16, cnt["4th-column test: v_bcnt_u32_b32"] / chunks / 16,
    cnt["register copies: v_mov_b32 / v_mov_b64"] / chunks / 16,
    (valu - core - cnt["Eq address: v_lshlrev_b32_sdwa (byte of the text dword -> LDS offset)"] - cnt["4th-column test: v_bcnt_u32_b32"] - cnt["register copies: v_mov_b32 / v_mov_b64"]) / chunks / 16))
if traffic:
    t = json.load(open(traffic))
    for sig, d in t.get("signatures", {}).items():
        st = d.get("stages", {}).get("mid_scan")
        if st and sig.startswith("ont:") and ":p=0:" in sig:
            print()
            print("PMC, one launch of the C2 kernel-path batch (%s; profiles/traffic.json, kernel sources %s):" % (sig, t.get("kernel_source_hash")))
            print("  SQ_INSTS_VALU = %.4e wave-instructions" % st["valu_insts_per_batch"])
            print("  (divide by the wave-columns of the launch -- bench.py's batch: the columns of the surviving reads' middle windows / 64 -- for the")
            print("   instructions per owned column pair: the loop's figure above x the warm-up columns of every stretch (80 per 4 096), the lanes")
            print("   parked or beyond the end of their wave's last stretch, the chunk that ends a window, the rare paths)")

# ---- the scan as a filter: k_mid_flat<2, Hot32, FS> (the last 32 rows of two adapters of 33..64 bp in the dword column) ----
def hot_loop(kernel):
    st = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    en = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
    b = lines[st:en + 1]
    pick = None
    for i, l in enumerate(b):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if not m or "Inner Loop Header" not in "\n".join(b[i:i + 4]):
            continue
        back = [j for j in range(i + 1, len(b)) if re.search(r"\bs_c?branch\w*\s+" + re.escape(m.group(1)) + r"$", b[j].strip())]
        if not back:
            continue
        reg = b[i:back[-1] + 1]
        n = sum(1 for x in reg if "v_bitop3_b32" in x)
        if pick is None or n > pick[0]:
            pick = (n, m.group(1), reg)
    return pick
print()
print("The scan as a filter (adapters of 33..64 bp within <= 12 differences): k_mid_flat<2, tgsf::Hot32, FS>, the adapter's last 32 rows in the")
print("10-instruction dword column (6 v_bitop3, v_and, v_add, 2 v_lshlrev per adapter), looked at every FS-th column (2 v_bcnt + 1 v_cmp per adapter)")
for fs in (1, 2):
    n, lab2, reg = hot_loop("_ZN4tgsf10k_mid_flatILi2ENS_5Hot32ELi%dEEEvNS_9DevParamsENS_8DevBatchEii" % fs)
    ins2 = [x.strip().split()[0] for x in reg if x.startswith("\t") and not x.strip().startswith((";", "."))]
    ch = n / 192.0                                  # 6 v_bitop3 per adapter and column: 192 per 16-column chunk of two adapters
    c2 = collections.Counter(ins2)
    v2 = sum(v for k, v in c2.items() if k.startswith("v_"))
    print("  FS = %d: inner loop %s, %d instructions a trip = %.0f chunks; VALU per column pair %.2f (the 64-bit column's loop above: %.2f): " % (fs, lab2, len(ins2), ch, v2 / ch / 16, valu / chunks / 16)
          + ", ".join("%s %.2f" % (k, v / ch / 16) for k, v in c2.most_common(9) if k.startswith(("v_", "ds_"))))

#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag> [bench.py args, e.g. --samples 2048 --ascans 1024 --bscans 512 --volumes 2 --out-slots 2]
# Produces under gpurun_out/<tag>/ :
#   kernel_stats.csv      rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 50 --warmup 5`
#   bench_in_profile.json the bench.py line of that same command (its HIP-event kernel time must agree)
#   pmc_counters.csv      pass,kernel,counter,dispatches,avg_value for separate --pmc passes
#   hbm_traffic.json      FETCH_SIZE / WRITE_SIZE -> HBM bytes per launch (gfx950 correction: FETCH_SIZE x 2)
# Every profiler run is wrapped in `timeout`; counters are collected with --kernel-trace only.
tag=$1; shift; root=$PWD; out=$root/gpurun_out/$tag
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 50 --warmup 5 --blocks 0 --no-cpu-baseline --no-extras --no-traffic "$@" > $out/bench_in_profile.json 2> $out/stats.log
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
i=0
echo "pass,kernel,counter,dispatches,avg_value" > $out/pmc_counters.csv
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 $root/bench.py --steps 8 --warmup 2 --warmup-seconds 0 --blocks 0 --no-cpu-baseline --no-extras --no-traffic "$@" > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "pass$i" >> $out/pmc_counters.csv <<'PY'
import csv, collections, sys
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if "oct" in k:
        print('%s,"%s",%s,%d,%g' % (sys.argv[2], k, c, len(v), sum(v) / len(v)))
PY
  rm -rf $out/p$i
done
rm -rf $out/stats
python3 - $out <<'PY'
import csv, json, sys
out = sys.argv[1]
# the kernel the traffic record is for must be the one bench.py names in roofline.kernel: fail loudly otherwise
bench = json.loads(open(out + "/bench_in_profile.json").read().strip().splitlines()[-1])
bench_kernel = bench["roofline"]["kernel"]
workload = "%dx%dx%d" % (bench["config"]["samples_per_ascan"], bench["config"]["ascans_per_bscan"], bench["config"]["bscans_per_buffer"])
f = w = None; name = None
for r in csv.DictReader(open(out + "/pmc_counters.csv")):
    if bench_kernel in r["kernel"]:
        name = r["kernel"]
        if r["counter"] == "FETCH_SIZE": f = float(r["avg_value"])
        if r["counter"] == "WRITE_SIZE": w = float(r["avg_value"])
if f and w:
    json.dump({"kernel": name.replace("void oct::", "").replace("(oct::FusedArgs)", ""), "workload": workload, "fetch_size_kb": f, "write_size_kb": w,
               "hbm_bytes_per_launch": (2 * f + w) * 1024.0,
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (values in KB); gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128 B request on coalesced streams -> doubled",
               "source": "pmc_counters.csv of the same run"}, open(out + "/hbm_traffic.json", "w"), indent=1)
else:
    sys.exit("profile_round.sh: no FETCH_SIZE / WRITE_SIZE rows for roofline.kernel = %r in pmc_counters.csv -- hbm_traffic.json NOT refreshed" % bench_kernel)
print(open(out + "/kernel_stats.csv").read().splitlines()[1][:160])
print(open(out + "/bench_in_profile.json").read().strip().splitlines()[-1][:400])
PY

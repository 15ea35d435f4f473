#!/bin/bash
# rocprofv3 passes over the bench workload; summaries land in gpurun_out/prof_<tag>/
tag=${1:-r01}; shift
args=${@:-"--steps 20 --warmup 5 --no-cpu-baseline --no-creep-leg"}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
# the chip's issue rate per instruction class (what the step kernel's instruction counts are priced against)
[ -x tools/valu_ceiling ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/valu_ceiling tools/valu_ceiling.hip
timeout -k 10 200 tools/valu_ceiling > $out/valu_ceiling.json || echo "valu_ceiling failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o trace --output-format csv -- python3 bench.py $args > $out/trace_bench.json 2> $out/trace.err || { echo trace failed; tail -5 $out/trace.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -d $out/pmc1 -o pmc1 --output-format csv -- python3 bench.py $args > /dev/null 2> $out/pmc1.err || { echo pmc1 failed; tail -5 $out/pmc1.err; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INST_LEVEL_LDS -d $out/pmc2 -o pmc2 --output-format csv -- python3 bench.py $args > /dev/null 2> $out/pmc2.err || { echo pmc2 failed; tail -5 $out/pmc2.err; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $out/pmc3 -o pmc3 --output-format csv -- python3 bench.py $args > /dev/null 2> $out/pmc3.err || { echo pmc3 failed; tail -5 $out/pmc3.err; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $out/pmc4 -o pmc4 --output-format csv -- python3 bench.py $args > /dev/null 2> $out/pmc4.err || { echo pmc4 failed; tail -5 $out/pmc4.err; }
find $out -name "*.csv" | head -20
python3 - <<PY
import csv, glob, collections, os
out="$out"
k4={}
for f in glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:3000])
for tag in ("pmc1","pmc2","pmc3","pmc4"):
    for f in glob.glob(out+f"/{tag}/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:40]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
            cnt[(k,r["Counter_Name"])]+=1
        with open(out+f"/{tag}_summary.txt","w") as w:
            for k,v in agg.items():
                for c,val in v.items():
                    line=f"{k:42s} {c:24s} per-launch {val/cnt[(k,c)]:.4g} launches {cnt[(k,c)]}"
                    print(line); w.write(line+"\n")
                    if "k_step_tiled" in k: k4[c]=val/cnt[(k,c)]
# the cache bench.py reads (roofline.traffic / valu_issue_frac), keyed to sources + workload
import json
line=json.loads(open(out+"/trace_bench.json").read().strip().splitlines()[-1])
fetch, write = k4.get("FETCH_SIZE"), k4.get("WRITE_SIZE")
summary={"command": "python bench.py $args (rocprofv3 --pmc, separate passes; tools/rocprof_passes.sh $tag)",
         "profile_key": line["config"]["profile_key"], "workload": line["config"]["workload"],
         "kernel": "k_step_tiled",
         "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write, "fetch_correction": 2.0,
         "traffic_bytes_per_launch": int((2.0*fetch+write)*1024) if fetch and write else None,
         "valu_wave_insts_per_launch": k4.get("SQ_INSTS_VALU"), "salu_wave_insts_per_launch": k4.get("SQ_INSTS_SALU"),
         "lds_wave_insts_per_launch": k4.get("SQ_INSTS_LDS"), "wave_quad_cycles_per_launch": k4.get("SQ_WAVE_CYCLES"),
         "valu_active_quad_cycles_per_launch": k4.get("SQ_ACTIVE_INST_VALU"),
         "lds_bank_conflict_cycles_per_launch": k4.get("SQ_LDS_BANK_CONFLICT"),
         "kernel_ms_in_this_run": line["roofline"]["kernel_ms"]}
json.dump(summary, open(out+"/k4_traffic.json","w"), indent=1)
print(json.dumps(summary))
PY

#!/bin/bash
# usage (under gpurun, after `bash tools/build_variant.sh trips -DCS_TILE_TRIPS` here): tools/refresh_profiles.sh <tag>
# Everything profiles/<tag>/ holds about the final sources, in one go: the kernel trace and the four PMC passes
# (tools/rocprof_passes.sh), the scene statistics from the trip counters, the bench table, the small-crowd legs, the
# kernel list of a 125k-agent step, and LAST the plain `python bench.py` line, which then finds the caches it quotes.
# The result is left under gpurun_out/profiles_<tag>/ (gpurun merges it back): copy it over profiles/<tag>/ and commit.
tag=${1:-r05}
cd $GRAFT_REPO_ROOT
dst=profiles/$tag
mkdir -p $dst gpurun_out
bash tools/rocprof_passes.sh $tag > gpurun_out/passes_$tag.log 2>&1 || { tail -5 gpurun_out/passes_$tag.log; exit 1; }
src=gpurun_out/prof_$tag
cp $(find $src/trace -name "*kernel_stats.csv" | head -1) $dst/kernel_stats.csv
for k in pmc1 pmc2 pmc3 pmc4; do cp $src/${k}_summary.txt $dst/; done
cp $src/k4_traffic.json $src/valu_ceiling.json $dst/
cp $src/trace_bench.json $dst/bench_under_trace.json
for w in walk creep random; do
  CS_LIB_PATH=$PWD/rmf_crowdsim_amd/lib/variants/trips.so timeout -k 10 200 python tools/trip_counts.py $w 1000000 > gpurun_out/trips_$w.txt 2>&1 || tail -3 gpurun_out/trips_$w.txt
done
python3 - <<PY
import json
out = {}
for w in ("walk", "creep", "random"):
    for line in open(f"gpurun_out/trips_{w}.txt"):
        if line.startswith("SCENE_STATS "):
            out.update(json.loads(line[len("SCENE_STATS "):]))
json.dump(out, open("$dst/scene_stats.json", "w"), indent=1)
print("scene statistics for", sorted(v["workload"] for v in out.values()))
PY
bash tools/bench_table.sh $tag > $dst/bench_table.txt 2>&1; cat $dst/bench_table.txt
bash tools/keep_bench.sh 125000 62500 > $dst/kept_windows_in_kernel_builder.txt 2>&1
bash tools/trace_small.sh > $dst/trace_small_125k.txt 2>&1; cat $dst/trace_small_125k.txt
# round 5's two priced levers, on the same box as everything else
timeout -k 10 300 python tools/sort_every_bench.py > $dst/sort_every.txt 2>&1; cat $dst/sort_every.txt
for v in 0 1; do for n in 1000000 125000; do CS_DEBUG_CTX_BY_VALUE=$v timeout -k 10 200 python bench.py --agents $n --steps 200 --no-cpu-baseline --no-creep-leg 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('ctx_by_value $v agents $n ms/step %.4f k4_ms %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"; done; done > $dst/ctx_by_value.txt 2>&1; cat $dst/ctx_by_value.txt
timeout -k 10 400 python bench.py > $dst/bench_default.json 2> gpurun_out/default_$tag.err || tail -3 gpurun_out/default_$tag.err
python3 -c "
import json; r=json.loads(open('$dst/bench_default.json').read().strip().splitlines()[-1])
print('default line: value %.4g ms/step %.4f k4 %.4f traffic %s valu_issue_frac %s scene_stats from %s' % (r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['traffic'], r['roofline']['valu_issue_frac'], r['scene_stats'].get('source')))"
rm -rf gpurun_out/profiles_$tag; mkdir -p gpurun_out/profiles_$tag
cp $dst/kernel_stats.csv $dst/pmc*_summary.txt $dst/k4_traffic.json $dst/valu_ceiling.json $dst/bench_under_trace.json $dst/scene_stats.json \
   $dst/bench_table.txt $dst/kept_windows_in_kernel_builder.txt $dst/trace_small_125k.txt $dst/bench_default.json \
   $dst/sort_every.txt $dst/ctx_by_value.txt gpurun_out/profiles_$tag/

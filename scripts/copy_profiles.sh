#!/bin/bash
# Copy the evidence of scripts/prof_round.sh TAG (gpurun_out/prof_TAG/) into profiles/ under the names profiles/README.md lists: scripts/copy_profiles.sh TAG [PREFIX]
# (PREFIX: the files' prefix in profiles/ when it differs from the run's tag, e.g. `copy_profiles.sh r4b r04`)
set -eu
SRC=${1:?tag}
TAG=${2:-$SRC}
S=gpurun_out/prof_$SRC; D=profiles
cpj() { [ -s "$S/$1" ] && tail -1 "$S/$1" > "$D/${TAG}_$2"; }
cpj bench_default.json bench_10k_x_1mib.json
for x in aes_ctr aes_gcm level1 level2 level7 level19; do cpj bench_$x.json bench_10k_x_1mib_$x.json; done
cpj bench_solid.json bench_solid_8192_x_1mib.json
cpj bench_deflate.json bench_deflate_2048_x_1mib.json
cpj bench_deflate_level1.json bench_deflate_2048_x_1mib_level1.json
cpj bench_deflate_level9.json bench_deflate_2048_x_1mib_level9.json
cpj bench_deflate_4k.json bench_deflate_262144_x_4kib.json
cpj bench_deflate_4k_1m.json bench_deflate_1000000_x_4kib.json
cpj bench_deflate_4k_text.json bench_deflate_262144_x_4kib_text.json
cpj bench_zstd_4k.json bench_zstd_262144_x_4kib.json
cpj bench_zstd_4k_1m.json bench_zstd_1000000_x_4kib.json
cpj bench_zstd_4k_text.json bench_zstd_262144_x_4kib_text.json
for k in default:10k_x_1mib deflate_4k_1m:deflate_1000000_x_4kib zstd_4k:zstd_262144_x_4kib; do
  f=$(find "$S/kt_${k%%:*}" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$D/${TAG}_kernel_stats_${k##*:}.csv"
done
for t in stream_rate batch_rate batch_sizes batch_latency host_rate zdec_one_frame decode_foreign_frames inflate_one_entry inflate_foreign_stream zdec_huge_frame inflate_huge_stream sq_counters; do [ -s "$S/$t.txt" ] && cp "$S/$t.txt" "$D/${TAG}_$t.txt"; done
[ -s "$S/host_rate_trace.txt" ] && cp "$S/host_rate_trace.txt" "$D/${TAG}_create_pipeline_trace.txt"
[ -s gpurun_out/pmc_sq/issue.json ] && cp gpurun_out/pmc_sq/issue.json "$D/${TAG}_sq_issue.json"
[ -s gpurun_out/pmc_tcc/summary.txt ] && cp gpurun_out/pmc_tcc/summary.txt "$D/${TAG}_lzp_tcc.txt"
[ -s "$S/pmc_summary_raw.json" ] && cp "$S/pmc_summary_raw.json" "$D/${TAG}_pmc_summary_raw.json"
ls "$D" | grep "^${TAG}_" | wc -l

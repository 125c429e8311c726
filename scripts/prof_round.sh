#!/bin/bash
# Round-end evidence on the GPU box: kernel-trace stats of the default bench, the bench lines of the named configurations next to them.
# Output under gpurun_out/prof_$TAG/ (TAG defaults to "i"); copy what is to be judged into profiles/.
set -u
export TMPDIR=/tmp
TAG=${1:-i}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
PART=${2:-all}
if [ "$PART" = all ] || [ "$PART" = a ]; then
python3 bench.py --steps 10 --warmup 3 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo default done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_default" -o kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > "$OUT/kt_default.log" 2>&1
echo trace done
python3 bench.py --framing solid --files 8192 > "$OUT/bench_solid.json" 2> "$OUT/bench_solid.err"
python3 bench.py --algo deflate --files 2048 > "$OUT/bench_deflate.json" 2> "$OUT/bench_deflate.err"
echo solid deflate done
python3 bench.py --no-cpu-baseline --no-end-to-end --encrypt aes-ctr > "$OUT/bench_aes_ctr.json" 2> "$OUT/bench_aes_ctr.err"
python3 bench.py --no-cpu-baseline --no-end-to-end --encrypt aes-gcm > "$OUT/bench_aes_gcm.json" 2> "$OUT/bench_aes_gcm.err"
for lv in 1 2 7 19; do python3 bench.py --no-cpu-baseline --no-end-to-end --level $lv > "$OUT/bench_level$lv.json" 2> "$OUT/bench_level$lv.err"; done
for lv in 1 9; do python3 bench.py --algo deflate --files 2048 --no-cpu-baseline --no-end-to-end --level $lv > "$OUT/bench_deflate_level$lv.json" 2> "$OUT/bench_deflate_level$lv.err"; done
echo levels done
fi
if [ "$PART" = all ] || [ "$PART" = c ]; then
# many small entries (BASELINE.json configs[4]'s shape and its zstd counterpart; kind 1 = random-text, kind 0 = enwik-style text)
python3 bench.py --algo deflate --files 262144 --file-mib 0.00390625 --kind 1 > "$OUT/bench_deflate_4k.json" 2> "$OUT/bench_deflate_4k.err"
python3 bench.py --algo deflate --files 1000000 --file-mib 0.00390625 --kind 1 > "$OUT/bench_deflate_4k_1m.json" 2> "$OUT/bench_deflate_4k_1m.err"
python3 bench.py --algo deflate --files 262144 --file-mib 0.00390625 --kind 0 --no-cpu-baseline --no-end-to-end > "$OUT/bench_deflate_4k_text.json" 2> "$OUT/bench_deflate_4k_text.err"
echo deflate small done
python3 bench.py --algo zstd --files 262144 --file-mib 0.00390625 --kind 1 > "$OUT/bench_zstd_4k.json" 2> "$OUT/bench_zstd_4k.err"
python3 bench.py --algo zstd --files 1000000 --file-mib 0.00390625 --kind 1 > "$OUT/bench_zstd_4k_1m.json" 2> "$OUT/bench_zstd_4k_1m.err"
python3 bench.py --algo zstd --files 262144 --file-mib 0.00390625 --kind 0 --no-cpu-baseline --no-end-to-end > "$OUT/bench_zstd_4k_text.json" 2> "$OUT/bench_zstd_4k_text.err"
echo zstd small done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_deflate_4k_1m" -o kt -- python3 bench.py --algo deflate --files 1000000 --file-mib 0.00390625 --kind 1 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-verify > "$OUT/kt_deflate_4k_1m.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_zstd_4k" -o kt -- python3 bench.py --files 262144 --file-mib 0.00390625 --kind 1 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-verify > "$OUT/kt_zstd_4k.log" 2>&1
echo small traces done
fi
if [ "$PART" = all ] || [ "$PART" = b ]; then
python3 scripts/stream_rate.py 4096 > "$OUT/stream_rate.txt" 2>&1
python3 scripts/batch_latency.py > "$OUT/batch_latency.txt" 2>&1
python3 scripts/batch_rate.py > "$OUT/batch_rate.txt" 2>&1
python3 scripts/batch_sizes.py > "$OUT/batch_sizes.txt" 2>&1
PNA_TRACE=1 python3 scripts/host_rate.py 10000 > "$OUT/host_rate.txt" 2> "$OUT/host_rate_trace.txt"
echo seam done
python3 scripts/zdec_one_frame_rate.py 256 > "$OUT/zdec_one_frame.txt" 2>&1
python3 scripts/zdec_one_frame_rate.py 1024 >> "$OUT/zdec_one_frame.txt" 2>&1
python3 scripts/decode_rate_foreign.py 2048 > "$OUT/decode_foreign_frames.txt" 2>&1
python3 scripts/inflate_one_entry_rate.py 1024 > "$OUT/inflate_one_entry.txt" 2>&1
python3 scripts/inflate_foreign_stream_rate.py 256 6 > "$OUT/inflate_foreign_stream.txt" 2>&1
python3 scripts/inflate_foreign_stream_rate.py 1024 6 >> "$OUT/inflate_foreign_stream.txt" 2>&1
echo decode done
bash scripts/pmc_insts.sh 4096 > "$OUT/sq_counters.txt" 2>&1
bash scripts/pmc_sq.sh 4096 >> "$OUT/sq_counters.txt" 2>&1
bash scripts/pmc_traffic.sh 10000 > "$OUT/pmc_traffic.log" 2>&1
cp gpurun_out/pmc/summary.json "$OUT/pmc_summary_raw.json" 2>/dev/null
find "$OUT" -name "*kernel_stats.csv" | head
fi
if [ "$PART" = all ] || [ "$PART" = d ]; then
# streams of 2 GiB and more: the parallel executor's windows (a failing parallel path is refused / falls to paths that end quickly: see the scripts)
for m in 2560 4200 6000; do timeout -k 10 200 python3 scripts/zdec_huge_frame.py $m >> "$OUT/zdec_huge_frame.txt" 2>&1; done
echo huge frames done
for m in 2200 4300; do timeout -k 10 400 python3 scripts/inflate_huge_stream.py $m 2>&1 | grep -v "  compressed" >> "$OUT/inflate_huge_stream.txt"; done
echo huge streams done
fi

set -e
for S in 0 1; do
  export PNA_LZ_SPLIT=$S PNA_LZ_SPLIT_BLOCKS=32768
  echo "== split=$S"
  python bench.py --steps 3 --warmup 1 --no-end-to-end --no-cpu-baseline --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('zstd 10k x 1MiB', d['value'], d['ms_per_step'], d.get('ratio'), d['roofline']['achieved'])"
  python bench.py --steps 3 --warmup 1 --algo deflate --files 2048 --no-end-to-end --no-cpu-baseline --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('deflate 2048 x 1MiB', d['value'], d['ms_per_step'])"
  python bench.py --steps 3 --warmup 1 --algo deflate --files 262144 --file-mib 0.00390625 --no-end-to-end --no-cpu-baseline --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('deflate 262144 x 4KiB', d['value'], d['ms_per_step'])"
  python bench.py --steps 3 --warmup 1 --level 19 --no-end-to-end --no-cpu-baseline --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('zstd L19 10k x 1MiB', d['value'], d['ms_per_step'])"
done

O=gpurun_out/profiles_r04h; mkdir -p $O
python bench.py > $O/r04h_bench.json 2> $O/bench.err
: > $O/r04h_grid.jsonl
for n in 256 1024 4096 16384; do for k in clustered uniform; do
  st=100; [ $n -ge 16384 ] && st=30
  python bench.py --boxes $n --kind $k --steps $st --warmup 10 --no-cpu-baseline --no-other-kind 2>/dev/null | tail -1 >> $O/r04h_grid.jsonl
done; done
for n in 4096 16384; do for k in clustered uniform; do
  st=100; [ $n -ge 16384 ] && st=30
  python bench.py --dim 3 --boxes $n --kind $k --steps $st --warmup 10 --no-cpu-baseline --no-other-kind 2>/dev/null | tail -1 >> $O/r04h_grid.jsonl
done; done
bash tools/prof.sh r04h_n16384 --boxes 16384 --steps 30 --warmup 5 --no-cpu-baseline --no-other-kind > $O/prof_n16384.txt 2>&1
cp gpurun_out/prof_r04h_n16384/bench_kernel_stats.csv $O/r04h_n16384_kernel_stats.csv
bash tools/prof.sh r04h_d3 --steps 30 --warmup 5 --no-cpu-baseline --no-other-kind --dim 3 --boxes 16384 > $O/prof_d3.txt 2>&1
cp gpurun_out/prof_r04h_d3/bench_kernel_stats.csv $O/r04h_dim3_n16384_kernel_stats.csv
bash tools/prof.sh r04h_d34k --steps 100 --warmup 5 --no-cpu-baseline --no-other-kind --dim 3 > $O/prof_d34k.txt 2>&1
cp gpurun_out/prof_r04h_d34k/bench_kernel_stats.csv $O/r04h_dim3_n4096_kernel_stats.csv
python tools/mode_times.py > $O/r04h_mode_times.jsonl 2>/dev/null

python bench.py --steps 5000 --warmup 10 > gpurun_out/_clk_bench.json 2>/dev/null &
BP=$!
for i in $(seq 1 60); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  rocm-smi --showclocks --showpower 2>&1 | grep -i -E "sclk|Socket" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 2
done
wait $BP
tail -1 gpurun_out/_clk_bench.json | cut -c1-200

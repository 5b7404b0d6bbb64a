for sh in "" "4,2" "2,4"; do
  if [ -n "$sh" ]; then export ACAS2D_SHAPE="$sh"; else unset ACAS2D_SHAPE; fi
  python - <<PY 2>/dev/null
import sys, os, torch, types
sys.path.insert(0, ".")
import bench, gym_acas2d_amd as g
args = bench.parse(["--no-cpu-baseline"])
for d in ("f64", "f64-fast"):
    o = bench.time_config(g, 65536, 8, d, torch.device("cuda", 0), args)
    print("shape %-5s %-9s launch %.2f us  %.3g env-steps/s  frac %.3f (G=%d C=%d)" % (os.environ.get("ACAS2D_SHAPE", "dflt"), d, o["launch_us"], o["env_steps_per_s"], o["frac"], o["lanes_per_env"], o["traffic_per_lane"]))
PY
done

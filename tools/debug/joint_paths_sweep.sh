cd $GRAFT_REPO_ROOT
for args in "--ns 8 --H 42 --iters 4" "--ns 8 --H 33 --iters 5" "--ns 8 --H 20 --iters 7" "--ns 8 --H 41 --iters 5" "--pendulum --ns 8 --H 25 --iters 7" "--pendulum --ns 8 --H 10 --iters 14" "--pendulum --ns 8 --H 42 --iters 4" "--ns 8 --H 40 --iters 5 --no-cache" "--ns 8 --H 12 --iters 12"; do
  echo "== $args"
  timeout 600 python tools/debug/joint_paths.py $args 2>&1 | grep "k=\|worst\|Error\|error" | awk '{ if ($0 ~ /worst/) print; else { n=split($0,a," "); print a[4], a[5], a[6], a[7], "mean", a[11], "cov", a[15], a[16], a[17], a[18], a[19], a[20], a[21] } }' | tail -16
done

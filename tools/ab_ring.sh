cd /root/repo
for pass in 1 2; do for name in default ring20 ring24; do
  if [ "$name" = default ]; then unset PTK_LIBRARY; else export PTK_LIBRARY=/root/repo/tools/bin/libptk_$name.so; fi
  for k in 16 4; do echo "== $name k=$k pass $pass"; timeout 300 python tools/ab_env.py --configs ";" --rounds 5 --k $k 2>&1 | tail -1 | cut -c1-260; done
  echo "== $name radius pass $pass"; timeout 300 python tools/ab_radius.py --configs ";" --rounds 3 2>&1 | tail -1
done; done

# Dev: A/B of library builds on the H=100 path.  usage: bash tools/ab_h100.sh lib1.so lib2.so ...
set -e
for f in "$@"; do
  echo "== $f"
  MPC_AMD_LIB=$PWD/$f timeout -k 10 150 python -u tools/quick_time.py 100 16384 2 2>&1 | grep "H=100"
done

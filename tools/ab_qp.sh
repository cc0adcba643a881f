for i in 1 2 3; do
  MPC_AMD_LIB=ab_libs/lib_base.so timeout -k 10 200 python -u tools/resident_time.py 65536 1 6 --no-devqp 2>&1 | grep resident | sed 's/^/base /'
  timeout -k 10 200 python -u tools/resident_time.py 65536 1 6 --no-devqp 2>&1 | grep resident | sed 's/^/new  /'
done

#!/bin/bash
# round 4, call 27: dispatch-threshold sweeps - the reduction over row lengths / row counts (three kernel routes), the ring
# kernels at every size (AUTO = mixed route up to N = 10, Jacobi above; ring_hh and jacobi forced for comparison)
R=$PWD; OUT=$R/gpurun_out/r4ai; mkdir -p $OUT
timeout -k 10 300 python scripts/reduce_sweep.py 2>&1 | grep -v amdgpu.ids | tee $OUT/reduce_sweep.txt
S10=$(python3 -c "print(','.join(f'{n}:100:10000' for n in range(3, 11)))")
S16=$(python3 -c "print(','.join(f'{n}:100:10000' for n in range(3, 17)))")
for k in auto ring_hh; do
  echo "== ring, kernel=$k" | tee -a $OUT/ring_sweep.txt
  timeout -k 10 300 python scripts/kbench.py --ring --kernel $k --reps 100 --shapes $S10 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ring_sweep.txt
done
echo "== ring, kernel=jacobi" | tee -a $OUT/ring_sweep.txt
timeout -k 10 400 python scripts/kbench.py --ring --kernel jacobi --reps 30 --shapes $S16 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ring_sweep.txt

# development aid: repeat the two-rank SyncBatchNorm test until it fails (at most 8 times)
for i in 1 2 3 4 5 6 7 8; do
  timeout -k 5 200 python -u -m pytest tests/test_data_parallel.py -q -x -k "sync_batchnorm" --timeout 190 --timeout-method thread > gpurun_out/sb_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc" >> gpurun_out/sb_summary.txt
  if [ $rc -ne 0 ]; then break; fi
done

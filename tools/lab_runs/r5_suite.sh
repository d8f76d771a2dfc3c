# the whole GPU suite + smoke on the final build
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/r5_gputests_final.log 2>&1
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2) > gpurun_out/r5_smoke_final.log 2>&1
cat gpurun_out/r5_gputests_final.log gpurun_out/r5_smoke_final.log

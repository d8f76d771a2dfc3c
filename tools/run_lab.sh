cd $GRAFT_REPO_ROOT
B=tools/bin
L=aphrodite_engine_amd/lib/libaphrodite_mi355x.so
mkdir -p gpurun_out
V=$B/lab_vf.so
timeout 200 $B/reslab --shapes gate_up,down,qkv $L "$V" "$V@APHRO_WNA16_RING=-8;RESLAB_TRACE=1" "$V@APHRO_WNA16_RING=-9" "$V@APHRO_WNA16_RING=-12;RESLAB_TRACE=1" \
  "$B/lab_vfq.so@APHRO_WNA16_RING=-8" "$B/lab_vfq.so@APHRO_WNA16_RING=-9" "$B/lab_vfq.so@APHRO_WNA16_RING=-12" \
  "$B/lab_vf3.so@APHRO_WNA16_RING=-8" "$B/lab_vf3.so@APHRO_WNA16_RING=-9" "$B/lab_vf3.so@APHRO_WNA16_RING=-12" "$V" \
  > gpurun_out/reslab7.jsonl 2> gpurun_out/reslab7.err
echo rc=$?
grep -v skipped gpurun_out/reslab7.jsonl | cut -c1-420
tail -5 gpurun_out/reslab7.err

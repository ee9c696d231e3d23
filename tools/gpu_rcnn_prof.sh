export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rcnn_prof -o r -- python $GRAFT_REPO_ROOT/tools/model_profile.py rcnn_r50c4 batch=1 fp16=1 > $GRAFT_REPO_ROOT/gpurun_out/rcnn_prof.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocprof_summary.py gpurun_out/rcnn_prof | head -30

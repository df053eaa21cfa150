cd $GRAFT_REPO_ROOT
bash profiles/run_profile.sh r05 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r05_run_profile.log 2>&1
bash profiles/train_pmc.sh r05 2 "3dmm rgb 3dmm_tuned rgb_tuned" > gpurun_out/r05_train_pmc.log 2>&1
bash profiles/step_trace.sh r05 2 8 "3dmm:tuned rgb:tuned 3dmm:frozen rgb:frozen" > gpurun_out/r05_step_trace.log 2>&1
ls gpurun_out/prof_r05 gpurun_out/r05_step_*.txt

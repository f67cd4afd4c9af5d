for a in 0 1 2 4 8 3 15; do
  export CSD_STEM_ABL=$a CSD_LIB_PATH=$GRAFT_REPO_ROOT/conditional_score_diffusion_amd/libcsd_hip_tune.so
  bash tools/timeline_run.sh fp16f8 stem$a; echo "abl=$a $(head -1 gpurun_out/timeline_stem$a.txt)"
done

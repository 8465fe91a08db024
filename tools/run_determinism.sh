OUT=gpurun_out/r6_det; mkdir -p $OUT
( TAG="ViT-L/4 batch 8 parity" CFG=large_4x4patch_2frames_1tube BATCH=8 CALLS=1500 timeout 900 python tools/determinism_check.py
  TAG="ViT-B/8 batch 32 parity" CFG=base_8x8patch_2frames_1tube BATCH=32 CALLS=3000 timeout 600 python tools/determinism_check.py
  TAG="ViT-B/8 batch 32 fast" MODE=fast CFG=base_8x8patch_2frames_1tube BATCH=32 CALLS=2000 timeout 600 python tools/determinism_check.py
  TAG="IMU model batch 16 parity" CFG=imu BATCH=16 CALLS=1000 timeout 900 python tools/determinism_check.py
  TAG="ViT-L/4 batch 8 parity, side stream" SIDE_STREAM=1 CFG=large_4x4patch_2frames_1tube BATCH=8 CALLS=500 timeout 600 python tools/determinism_check.py ) > $OUT/determinism.log 2>&1
grep -v amdgpu $OUT/determinism.log | tail -8

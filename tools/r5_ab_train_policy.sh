L=composer_amd/lib
python tools/ab_step.py $L/r4_baseline.so $L/libcomposer_hip.so $L/xc_r.so $L/xc_rp.so --rounds 2 --cfg c2,c2b32 2>&1 | tail -2

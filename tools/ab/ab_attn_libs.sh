for v in old new old new; do cp tools/ab/lib_$v.so tc_light_amd/libtclight_hip.so; echo == $v; python tools/micro/bench_attn.py 2>&1 | grep "d="; done
cp tools/ab/lib_new.so tc_light_amd/libtclight_hip.so

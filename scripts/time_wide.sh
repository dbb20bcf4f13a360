#!/bin/bash
# forward time of the C5-shard Koopman forward under ablation builds of the wide kernel (build_abl/libisk_wabl<N>.so)
python scripts/time_fwd_wide.py 5
for lib in $(ls build_abl/libisk_wabl*.so 2>/dev/null); do ISK_LIB=$lib python scripts/time_fwd_wide.py 5; done

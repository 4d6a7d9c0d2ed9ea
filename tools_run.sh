bash tools/profile_r01.sh r01

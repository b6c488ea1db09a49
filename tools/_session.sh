tools/ab_side.sh r05i_sc "product" 2

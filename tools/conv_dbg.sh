# timing experiments: which part of the kernel costs what (needs a library built with -DTRTX_CONV_ABLATE:
#   make -C tensorrtx_amd/csrc clean && make -C tensorrtx_amd/csrc COMMON_EXTRA=-DTRTX_CONV_ABLATE).  dbg bits: 1 A loads range-checked away, 2 B loads,
# 4 no ds_read/MFMA, 8 no epilogue, 16 no k-loop
for d in 0 7 15 23 31; do echo -n "dbg=$d "; TRTX_CONV_DBG=$d python tools/conv_ab.py d$d 0,1,4,23,25,27,30 2>&1 | tail -1; done

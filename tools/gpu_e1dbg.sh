for d in 0 1 2 4 3 6; do DFA_E1_DBG=$d python tools/gpu_cae_ab.py cae_enc1_mfma=1 2>/dev/null | grep "enc1" | tail -1 | sed "s/^/dbg=$d /"; done

# A/B of the fused decoder's tile size (latent pixels per workgroup): rebuild with -DDFA_CDF_NP=32 on the box and time again
set -e
python tools/gpu_cae_ab.py cae_dec_fused=1 | tail -2
python tools/gpu_cae_dec_stamps.py | tail -1
cd deep-fake-audio-classifier_amd/csrc && touch cae_dec_fused.hip && make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DDFA_CDF_NP=32" > /dev/null 2>&1 && cd ../..
echo "--- NP=32"
python tools/gpu_cae_ab.py cae_dec_fused=1 | tail -2
python tools/gpu_cae_dec_stamps.py | tail -1

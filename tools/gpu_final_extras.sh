mkdir -p gpurun_out/final
timeout 200 python tools/trace_k256m.py --hidden 8192 --out gpurun_out/final/trace_k256m_h8192_cold.json 2>&1 | grep -v amdgpu.ids > gpurun_out/final/trace_k256m_h8192.txt
timeout 200 python tools/trace_k256m.py --hidden 8192 --hot --out gpurun_out/final/trace_k256m_h8192_hot.json 2>&1 | grep -v amdgpu.ids >> gpurun_out/final/trace_k256m_h8192.txt
timeout 200 python tools/trace_k256m.py --kernel valu --hidden 8192 --out gpurun_out/final/trace_k256_valu_h8192_cold.json 2>&1 | grep -v amdgpu.ids >> gpurun_out/final/trace_k256m_h8192.txt
cat gpurun_out/final/trace_k256m_h8192.txt
timeout 900 python tools/llama_decode.py --out gpurun_out/final/llama8b_decode.json 2>&1 | tail -5

#!/bin/bash
# One metered probe of the GPU box: is there anything on it that would let the
# reference's own arithmetic (ONNX Runtime + model.onnx + voices) run beside ours?  Writes
# gpurun_out/probe_box.txt; the answer is recorded in DESIGN.md §4.
out=gpurun_out/probe_box.txt
mkdir -p gpurun_out
{
  echo "## date: $(date -u)"
  echo "## python modules"
  for m in onnxruntime onnx kokoro kokoro_onnx misaki espeakng_loader phonemizer; do
    python3 -c "import $m, sys; print('$m', getattr($m, '__version__', '?'), $m.__file__)" 2>&1 | tail -1
  done
  echo "## shared libraries named onnxruntime"
  find / -xdev \( -name 'libonnxruntime*' -o -name 'onnxruntime*' \) -not -path '/proc/*' 2>/dev/null | head -20
  echo "## model / voice files (size)"
  find / -xdev \( -name '*.onnx' -o -name 'voices*.bin' -o -name 'voices*.npz' -o -name 'kokoro*.pth' -o -name 'kokoro*.safetensors' -o -name 'af_sky*' \) \
       -not -path '/proc/*' -size +100k 2>/dev/null | head -40 | while read -r f; do ls -l "$f"; done
  echo "## HF cache"
  ls -la ~/.cache/huggingface 2>&1 | head -5
  ls -la /root/.cache/huggingface 2>&1 | head -5
  echo "## network"
  timeout 8 python3 - <<'EOF' 2>&1 | tail -3
import socket
for host in ("huggingface.co", "github.com", "pypi.org"):
    try:
        socket.setdefaulttimeout(3)
        ip = socket.gethostbyname(host)
        s = socket.create_connection((ip, 443), timeout=3)
        s.close()
        print(host, "reachable", ip)
    except Exception as e:  # noqa: BLE001
        print(host, "unreachable:", type(e).__name__, e)
EOF
  echo "## pip download attempt (no install)"
  timeout 20 python3 -m pip download --no-deps -d /tmp/_probe onnxruntime 2>&1 | tail -2
  echo "## host"
  nproc; free -g | head -2
  rocminfo 2>/dev/null | grep -E 'Marketing Name|gfx' | head -4
} > "$out" 2>&1
cat "$out"

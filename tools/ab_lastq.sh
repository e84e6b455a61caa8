for f in True False; do
echo "== LASTQ_FROM_X=$f"
python - <<PY 2>&1 | grep -E "^\[|passed|failed"
import sys
sys.path.insert(0, "tests")
import recguru_amd.ops as o
o.LASTQ_FROM_X = $f
import pytest
pytest.main(["tests/test_steps_gpu.py", "-m", "gpu", "-q", "-s", "-k", "large_batch_bf16 or bench_shape_steps or loss_curves_replay"])
PY
done

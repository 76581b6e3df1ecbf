timeout 900 python -m pytest tests/test_gpu_python_recipes.py tests/test_gpu_parity.py -q -m gpu -k "struct_size or tolerance or recipe or switches" 2>&1 | tail -3

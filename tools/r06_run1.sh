timeout 900 python -m pytest tests/test_gpu_python_recipes.py -q -m gpu 2>&1 | tail -3

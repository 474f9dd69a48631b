timeout 600 python -m pytest tests/test_headless_app.py -m gpu -q 2>&1 | tail -12

"""python tools/first_contact.py --model_path ... [--resume ...] --dataset ...   (= python -m blim_amd.first_contact; see its docstring)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blim_amd.first_contact import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())

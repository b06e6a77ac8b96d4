"""Constants the hot-path operators read (reference: src/pydrobert/torch/config.py)."""

INDEX_PAD_VALUE = -100  # config.py:55
DEFT_INS_COST = 1.0  # config.py:156
DEFT_DEL_COST = 1.0  # config.py:159
DEFT_SUB_COST = 1.0  # config.py:162
DEFT_PAD_VALUE = 0.0

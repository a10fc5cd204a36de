"""CPU oracle for the LVGL pixel path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(vvc_amd/) never does.
"""

"""CPU oracle -- test infrastructure only (see vtoonify_oracle.py header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (vtoonify_amd/) never does.
"""

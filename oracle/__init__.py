"""oracle/ -- CPU checker for the PNA compress path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (portable-network-archive_amd/) never does.
"""

"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the hnswlib-backed nearest-neighbour path of
featureform/embeddinghub (see hnsw_oracle.hpp for provenance and pinning status).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; nothing under embeddinghub_amd/ does.
"""

"""the product library and diagnostic builds on the same buffer in one process: what the instrumentation costs"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.system("cd %s && python3 tools/ab_libs.py 1000 256 - exp/libvhp_TL.so exp/libvhp_NWNM.so exp/libvhp_TL_NWNM.so exp/libvhp_NOSTORE.so exp/libvhp_TL_NOSTORE.so" % ROOT)

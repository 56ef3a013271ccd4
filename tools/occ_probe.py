"""Diagnostic (GPU box): which layout and residency the engine picks at B = 8192 (MPCQ_VERBOSE line); MPCQ_LIB names the build (tools/r6_ab.sh variants)."""
import os, sys
os.environ["MPCQ_TUNING"]="1"; os.environ["MPCQ_VERBOSE"]="1"
sys.path.insert(0,'.')
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
Engine(EngineConfig(batch=8192,N=20,quad=hummingbird(),nb=10,basis=rgp_basis_linspace(12.0,10)),lib_path=os.environ.get("MPCQ_LIB")).close()

"""deepcubea_amd — MI355X-native batched weighted-A* (BWAS) node-expansion path for DeepCubeA.

Only the hot path named in BASELINE.json's north_star lives here:
  csrc/            HIP kernels (gfx950) + the C ABI of include/dca.h  -> libdca_hip.so
  _lib.py          ctypes loader (fails loudly when the library or a GPU is missing)
  environments/    Environment API mirror (environment_abstract.py / cube3.py / n_puzzle.py), HIP-backed
  search_methods/  astar.py CLI mirror with --language hip (bwas_hip)
  utils/           nnet_utils / pytorch_models / env_utils / search_utils mirrors for that path
"""
__version__ = "0.1.0"

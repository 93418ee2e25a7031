"""Import-path shim: `SemGCN.utils` of the reference tree (PengfeiRen96/DIR) resolved to the dir_amd mirror.

  sys.path.insert(0, "<repo>/dir_amd/compat")

lets the reference's callers keep their import lines unchanged (apps/eval.py:15-19 `from models.dir import DIR`,
`from models.manolayer import ManoLayer`; models/dir.py:7-15).  Plumbing only: every name is re-exported from dir_amd."""
from dir_amd.SemGCN.utils import *  # noqa: F401,F403
from dir_amd.SemGCN.utils import adj_mx_from_edges, get_sketch_setting  # noqa: F401

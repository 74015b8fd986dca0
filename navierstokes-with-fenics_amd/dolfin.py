"""``import dolfin`` shim: the reference's demos and tests import dolfin for the handful of objects
that cross the solver boundary (``Expression``, ``Constant``, ``SubDomain``, ``near``,
``DOLFIN_EPS``, ``pi``, ``set_log_level``, ``info``, ``Function``, ``project`` ...).  With this
package directory on PYTHONPATH *instead of* a FEniCS installation they resolve to the dolfin-free
stand-ins of ``dlfn_compat`` / ``fem_spaces``, so ``python demo/cavity_flow.py`` of the reference
runs unchanged on the MI355X path.  (A real dolfin earlier on the path wins, as it should.)"""
from dlfn_compat import *  # noqa: F401,F403
import dlfn_compat as _compat


def __getattr__(name):
    return getattr(_compat, name)

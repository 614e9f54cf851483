"""Import alias for the package directory ``hirag-prod_amd`` (a hyphen is not a valid identifier).

``import hirag_prod_amd`` executes this file, which loads ``hirag-prod_amd/__init__.py`` as the
package ``hirag_prod_amd`` and puts it in ``sys.modules`` under that name, so that
``hirag_prod_amd.store`` and the other submodules resolve inside the hyphenated directory.
"""

import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hirag-prod_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)

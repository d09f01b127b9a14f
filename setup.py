"""Builds the Cython binding distributions_amd/_core (in place).

The HIP library itself is built by distributions_amd/csrc/Makefile; use
`python __graft_entry__.py` (or __graft_entry__.build()) to build both.
"""
import os

import numpy
from Cython.Build import cythonize
from setuptools import Extension, setup

HERE = os.path.dirname(os.path.abspath(__file__))

ext = Extension(
    "distributions_amd._core",
    sources=["distributions_amd/_core.pyx"],
    include_dirs=[os.path.join(HERE, "include"), numpy.get_include()],
    library_dirs=[os.path.join(HERE, "distributions_amd")],
    libraries=["distributions_hip"],
    runtime_library_dirs=["$ORIGIN"],
    define_macros=[("NPY_NO_DEPRECATED_API", "NPY_1_7_API_VERSION")],
    language="c",
)

setup(
    name="distributions_amd",
    version="0.1.0",
    packages=["distributions_amd", "distributions_amd.lp",
              "distributions_amd.lp.models"],
    ext_modules=cythonize([ext], language_level=3),
)

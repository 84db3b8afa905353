"""Import alias: the package directory is ``parquet-java_amd`` (not a valid Python identifier);
this shim makes it importable as ``parquet_java_amd``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "parquet-java_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))

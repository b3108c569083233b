"""Import alias.  The package lives in `guided-attention_amd/` (the layout this repo is asked to
have); a hyphen is not a legal Python identifier, so this stub makes it importable as
`guided_attention_amd` by pointing the package search path at that directory."""
import pathlib as _pathlib

_real = _pathlib.Path(__file__).resolve().parent.parent / "guided-attention_amd"
__path__ = [str(_real)]
exec(compile((_real / "__init__.py").read_text(), str(_real / "__init__.py"), "exec"))

"""Part of the `model` mirror package (see model/__init__.py): merged with the reference's directory."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

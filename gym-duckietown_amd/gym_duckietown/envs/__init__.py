from .duckietown_env import DuckietownEnv, DuckietownLF  # noqa: F401
from .multimap_env import MultiMapEnv  # noqa: F401

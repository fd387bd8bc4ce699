from .parallel_generator import generate_ti2ti  # noqa: F401

from .views import complex_view, fix_dim  # noqa: F401

"""Root pytest configuration: `pytest` run from the repository root collects tests/ only.  experiments/ holds rejected
variants whose extensions build() does not compile; their test files are not part of the product's suite."""
collect_ignore_glob = ["experiments/*", "tools/*", "gpurun_out/*"]

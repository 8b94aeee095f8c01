# Convenience wrapper: everything native, in-tree (same as `python -c "import __graft_entry__ as g; g.build()"`).
all:
	$(MAKE) -C sdf-viewer_amd/csrc
	$(MAKE) -C sdf-viewer_amd/host
	$(MAKE) -C oracle

test-cpu: all
	python -m pytest tests -x -q -m "not gpu"

test-gpu: all
	python -m pytest tests -x -q -m gpu

clean:
	$(MAKE) -C sdf-viewer_amd/csrc clean
	$(MAKE) -C sdf-viewer_amd/host clean
	$(MAKE) -C oracle clean
.PHONY: all test-cpu test-gpu clean

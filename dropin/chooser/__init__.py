"""`chooser` package shim: put this directory's parent (`dropin/`) and the repo
root on PYTHONPATH and Spearmint's drivers --
    importlib.import_module('chooser.' + options.chooser_module)
(spearmint/spearmint/main.py:164, spearmint-lite/spearmint-lite.py:99) -- load the
MI355X choosers under the reference's own module names, so `--method=GPEIOptChooser`
and the `<module>.pkl` state files keep their names."""

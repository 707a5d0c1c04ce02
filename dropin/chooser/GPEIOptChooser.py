"""Drop-in `chooser.GPEIOptChooser`: same module name, same `init`/`next`, same
`chooser.GPEIOptChooser.pkl` state file as the reference module it shadows
(spearmint/spearmint/chooser/GPEIOptChooser.py); the EI grid runs on the GPU via libspx.so."""
from spearmint_amd import util as _util
from spearmint_amd.chooser import GPEIOptChooser as _impl


class GPEIOptChooser(_impl.GPEIOptChooser):
    # defined here so that self.__module__ == "chooser.GPEIOptChooser", which names the
    # state pickle exactly as the reference does (GPEIOptChooser.py: state_pkl)
    pass


def init(expt_dir, arg_string):
    return GPEIOptChooser(expt_dir, **_util.unpack_args(arg_string))

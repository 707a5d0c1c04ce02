"""Drop-in `chooser.GPEIperSecChooser`: same module name, same `init`/`next`, same
`chooser.GPEIperSecChooser.pkl` state file as the reference module it shadows
(spearmint/spearmint/chooser/GPEIperSecChooser.py); the EI grid runs on the GPU via libspx.so."""
from spearmint_amd import util as _util
from spearmint_amd.chooser import GPEIperSecChooser as _impl


class GPEIperSecChooser(_impl.GPEIperSecChooser):
    # defined here so that self.__module__ == "chooser.GPEIperSecChooser", which names the
    # state pickle exactly as the reference does (GPEIperSecChooser.py: state_pkl)
    pass


def init(expt_dir, arg_string):
    return GPEIperSecChooser(expt_dir, **_util.unpack_args(arg_string))

from .qcnet_sliding import QCNetSliding
from .temp_ens_learn_agg import TempEnsLearnAgg

__all__ = ['QCNetSliding', 'TempEnsLearnAgg']

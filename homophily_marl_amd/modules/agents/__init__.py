from .homophily_agent import HomophilyAgent

REGISTRY = {"homophily": HomophilyAgent}
